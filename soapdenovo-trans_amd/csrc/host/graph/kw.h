/* kw.h -- host-side k-mer words for the graph phases (after the GPU hashing pass).
 * A k-mer is always carried in four 64-bit words, most significant first (= the reference's MER127 struct
 * order high1,low1,high2,low2, inc/def.h:45-59); variants with fewer words use the tail.  Semantics of every
 * helper follow kmer.c (createFilter :313, nextKmer :209, prevKmer :230, reverseComplement :548-656,
 * KmerSmaller :27, first/lastCharInKmer :267-311). */
#ifndef SDT_KW_H
#define SDT_KW_H
#include <stdint.h>

typedef struct { uint64_t w[4]; } kw_t;

static inline int kw_less(const kw_t *a, const kw_t *b)
{
	for (int i = 0; i < 4; i++)
		if (a->w[i] != b->w[i]) return a->w[i] < b->w[i];
	return 0;
}
static inline int kw_eq(const kw_t *a, const kw_t *b)
{
	return a->w[0] == b->w[0] && a->w[1] == b->w[1] && a->w[2] == b->w[2] && a->w[3] == b->w[3];
}
/* bit i (0 = least significant of w[3]) helpers on the 256-bit value */
static inline unsigned kw_get2(const kw_t *k, int bit) { return (unsigned)(k->w[3 - (bit >> 6)] >> (bit & 63)) & 3u; }
static inline void kw_or2(kw_t *k, int bit, unsigned v) { k->w[3 - (bit >> 6)] |= (uint64_t)v << (bit & 63); }

static inline kw_t kw_mask(int K)
{
	kw_t f = {{0, 0, 0, 0}};
	int bits = 2 * K;
	for (int i = 3; i >= 0 && bits > 0; i--, bits -= 64)
		f.w[i] = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
	return f;
}
/* append base b at the low end, drop the first base */
static inline kw_t kw_next(kw_t k, unsigned b, int K)
{
	const kw_t m = kw_mask(K);
	kw_t r;
	r.w[0] = ((k.w[0] << 2) | (k.w[1] >> 62)) & m.w[0];
	r.w[1] = ((k.w[1] << 2) | (k.w[2] >> 62)) & m.w[1];
	r.w[2] = ((k.w[2] << 2) | (k.w[3] >> 62)) & m.w[2];
	r.w[3] = ((k.w[3] << 2) & m.w[3]) | b;
	return r;
}
/* prepend base b at the high end, drop the last base */
static inline kw_t kw_prev(kw_t k, unsigned b, int K)
{
	kw_t r;
	r.w[3] = (k.w[3] >> 2) | (k.w[2] << 62);
	r.w[2] = (k.w[2] >> 2) | (k.w[1] << 62);
	r.w[1] = (k.w[1] >> 2) | (k.w[0] << 62);
	r.w[0] = k.w[0] >> 2;
	kw_or2(&r, 2 * (K - 1), b);
	return r;
}
static inline uint64_t kw_rev2(uint64_t x)
{
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	return __builtin_bswap64(x);
}
static inline kw_t kw_rc(kw_t k, int K)
{
	uint64_t t[4];
	for (int i = 0; i < 4; i++)
		t[i] = kw_rev2(k.w[3 - i] ^ 0xAAAAAAAAAAAAAAAAULL);
	/* right-align: shift the 256-bit value right by 256 - 2K */
	const int s = 256 - 2 * K, ws = s >> 6, bs = s & 63;
	kw_t r;
	for (int i = 3; i >= 0; i--) {
		const int src = i - ws;
		uint64_t v = 0;
		if (src >= 0) {
			v = t[src] >> bs;
			if (bs && src >= 1) v |= t[src - 1] << (64 - bs);
		}
		r.w[i] = v;
	}
	return r;
}
/* reverseComplement(x, K + 1) as the reference computes it for the (K+1)-mers of length-1 edges (node2edge.c:404-463,
 * prlRead2path.c:719-738): fastReverseComp takes the length as a `char` (kmer.c:548), so 128 (K = 127) arrives as
 * -128, takes the "shorter than 32" branch and only complements + reverses the LAST word (its shift count, 320, acts
 * as 0 on x86-64).  The 127mer binary's patch table and its look-ups both live with that; so do we. */
static inline kw_t kw_rc_kplus1(kw_t plus, int K)
{
	if (K + 1 < 128) return kw_rc(plus, K + 1);
	kw_t r = plus;
	r.w[3] = kw_rev2(plus.w[3] ^ 0xAAAAAAAAAAAAAAAAULL);
	return r;
}
static inline unsigned kw_first(const kw_t *k, int K) { return kw_get2(k, 2 * (K - 1)); }
static inline unsigned kw_last(const kw_t *k) { return (unsigned)k->w[3] & 3u; }

#endif
