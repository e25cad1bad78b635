/*
 * sdt_oracle.c -- CPU restatement of the SOAPdenovo-Trans `pregraph` hashing path.
 *
 * TEST INFRASTRUCTURE ONLY (see sdt_oracle.h).  Single-threaded, written for clarity; every function
 * cites the reference file:line it follows (paths relative to /root/reference/src).
 * Parity of this file against the reference is pinned by tests/test_oracle_vs_reference.py
 * (unit vectors from the reference's own objects + *.kmerFreq files written by the reference binary).
 */
#include "sdt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ base coding */

/* inc/def.h:39  base2int(base) = ((base)&0x06)>>1 : A->0 C->1 T->2 G->3 (and N->3) */
int sdto_base2int(int c) { return (c & 0x06) >> 1; }

/* readseq1by1.c:296-326 (readseqfq) / :150-173 (readseqInBuf): the raw line is cut to max_read_len
 * characters first; then lowercase letters are folded, letters are coded with base2int, '.' becomes A,
 * everything else is dropped.  (N_kmer / -n path is off: survey 9.3-q11.) */
int sdto_encode_line(const char *str, int strL, int max_read_len, uint8_t *codes)
{
	int n = 0;
	if (strL > max_read_len)
		strL = max_read_len;
	for (int i = 0; i < strL; i++) {
		char ch = str[i];
		if (ch >= 'a' && ch <= 'z')
			codes[n++] = (uint8_t)sdto_base2int(ch - 'a' + 'A');
		else if (ch >= 'A' && ch <= 'Z')
			codes[n++] = (uint8_t)sdto_base2int(ch);
		else if (ch == '.')
			codes[n++] = (uint8_t)sdto_base2int('A');
	}
	return n;
}

/* ------------------------------------------------------------------ Kmer arithmetic */

static sdto_kmer k_zero(void) { sdto_kmer z = {{0, 0, 0, 0}}; return z; }

/* kmer.c:150-168 KmerLeftBitMoveBy2 */
static sdto_kmer k_shl2(sdto_kmer k)
{
	k.w[0] = (k.w[0] << 2) | (k.w[1] >> 62);
	k.w[1] = (k.w[1] << 2) | (k.w[2] >> 62);
	k.w[2] = (k.w[2] << 2) | (k.w[3] >> 62);
	k.w[3] <<= 2;
	return k;
}

/* kmer.c:171-189 KmerRightBitMoveBy2 */
static sdto_kmer k_shr2(sdto_kmer k)
{
	k.w[3] = (k.w[3] >> 2) | (k.w[2] << 62);
	k.w[2] = (k.w[2] >> 2) | (k.w[1] << 62);
	k.w[1] = (k.w[1] >> 2) | (k.w[0] << 62);
	k.w[0] >>= 2;
	return k;
}

/* kmer.c:357-407 KmerRightBitMove, any distance 0..255 */
static sdto_kmer k_shr(sdto_kmer k, int dis)
{
	sdto_kmer r = k_zero();
	int ws = dis >> 6, bs = dis & 63;
	for (int i = 3; i >= 0; i--) {
		int src = i - ws;
		uint64_t v = 0;
		if (src >= 0) {
			v = k.w[src] >> bs;
			if (bs && src - 1 >= 0)
				v |= k.w[src - 1] << (64 - bs);
		}
		r.w[i] = v;
	}
	return r;
}

/* kmer.c:313-355 createFilter: 4^K - 1 */
sdto_kmer sdto_create_filter(int K)
{
	sdto_kmer f = k_zero();
	int bits = 2 * K;
	for (int i = 3; i >= 0 && bits > 0; i--) {
		f.w[i] = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
		bits -= 64;
	}
	return f;
}

/* kmer.c:209-228 nextKmer: shift a base in at the low end, mask with WORDFILTER */
sdto_kmer sdto_next_kmer(sdto_kmer prev, int ch, int K)
{
	sdto_kmer f = sdto_create_filter(K);
	sdto_kmer w = k_shl2(prev);
	for (int i = 0; i < 4; i++)
		w.w[i] &= f.w[i];
	w.w[3] |= (uint64_t)ch;
	return w;
}

/* kmer.c:230-265 prevKmer: shift a base in at bit 2(K-1) */
sdto_kmer sdto_prev_kmer(sdto_kmer next, int ch, int K)
{
	sdto_kmer w = k_shr2(next);
	int bit = 2 * (K - 1);
	w.w[3 - (bit >> 6)] |= (uint64_t)ch << (bit & 63);
	return w;
}

/* kmer.c:267-311 */
int sdto_last_char(sdto_kmer k) { return (int)(k.w[3] & 3); }
int sdto_first_char(sdto_kmer k, int K)
{
	int bit = 2 * (K - 1);
	/* the reference returns the (char)-truncated shifted word without masking; for a filtered k-mer
	 * only 2 bits remain, so the mask is a no-op there */
	return (int)((k.w[3 - (bit >> 6)] >> (bit & 63)) & 3);
}

/* kmer.c:27-128 */
int sdto_kmer_smaller(sdto_kmer a, sdto_kmer b)
{
	for (int i = 0; i < 4; i++)
		if (a.w[i] != b.w[i])
			return a.w[i] < b.w[i];
	return 0;
}
int sdto_kmer_equal(sdto_kmer a, sdto_kmer b)
{
	return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}

/* swap the 2-bit groups of a word end for end (the 5-stage network of kmer.c:629-645, MER31/63) */
static uint64_t rev2bit(uint64_t x)
{
	x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xCCCCCCCCCCCCCCCCULL) >> 2);
	x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
	x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
	x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
	return (x << 32) | (x >> 32);
}

/* kmer.c:548-656 fastReverseComp/reverseComplement: complement every base (^0b10), reverse the order
 * of the 2-bit groups over the whole register, then right-align the 2K significant bits. */
/* KmerPlus (kmer.c:191-207): append a base WITHOUT masking -- the (K+1)-mer of a length-1 edge */
sdto_kmer sdto_kmer_plus(sdto_kmer prev, int ch)
{
	sdto_kmer w = k_shl2(prev);
	w.w[3] |= (uint64_t)(ch & 3);
	return w;
}

sdto_kmer sdto_reverse_complement(sdto_kmer k, int K)
{
	sdto_kmer r;
	for (int i = 0; i < 4; i++)
		r.w[i] = rev2bit(k.w[3 - i] ^ 0xAAAAAAAAAAAAAAAAULL);
	return k_shr(r, 256 - 2 * K);
}

/* reverseComplement(wordplus, overlaplen + 1) as the reference computes it.  fastReverseComp takes the length as a `char`
 * (kmer.c:548): with the 127mer binary and K = 127 the 128 bases arrive as -128, take the "shorter than 32 bases" branch
 * (:553-557) and only the LAST word is complemented and reversed (the shift count 64 - (-256) is 0 mod 64 on x86); every
 * other length is the ordinary reverse complement. */
sdto_kmer sdto_rc_kplus1(sdto_kmer wordplus, int K, int nw)
{
	if (nw == 4 && K + 1 == 128) {
		sdto_kmer r = wordplus;
		r.w[3] = rev2bit(wordplus.w[3] ^ 0xAAAAAAAAAAAAAAAAULL);
		return r;
	}
	return sdto_reverse_complement(wordplus, K + 1);
}

/* ------------------------------------------------------------------ owner hash */

/* hashFunction.c:28-81 holds the standard reflected CRC-32 table (poly 0xEDB88320) as `int`s;
 * regenerate it instead of copying 256 literals. */
static int32_t crc_table[256];
static int crc_ready;
static void crc_init(void)
{
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int k = 0; k < 8; k++)
			c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		crc_table[n] = (int32_t)c;
	}
	crc_ready = 1;
}

/* hashFunction.c:83-98 crc32(): state is a *signed int* and `crc >> 8` is an arithmetic shift, so the
 * sign bit smears into the top byte -- this is NOT zlib's CRC-32 (survey 9.4). buf is `const char*`
 * (signed) but only its low 8 bits survive the & 0xff. */
static int32_t ref_crc32(int32_t crc, const unsigned char *buf, int len)
{
	crc = crc ^ (int32_t)0xffffffff;
	while (len--)
		crc = crc_table[(crc ^ (int32_t)(signed char)*buf++) & 0xff] ^ (crc >> 8);
	return crc ^ (int32_t)0xffffffff;
}

/* hashFunction.c:108-122 hash_kmer: CRC over sizeof(Kmer) raw little-endian bytes of the variant's
 * struct ({high1,low1,high2,low2} / {high,low} / scalar), sign-extended into ubyte8, & 0xffffff */
uint64_t sdto_hash_kmer(sdto_kmer k, int nw)
{
	if (!crc_ready)
		crc_init();
	uint64_t raw[4];
	for (int i = 0; i < nw; i++)
		raw[i] = k.w[4 - nw + i];
	int32_t c = ref_crc32(0, (const unsigned char *)raw, 8 * nw);
	return (uint64_t)(int64_t)c & 0xffffffULL;
}

/* ------------------------------------------------------------------ chopKmer4read */

/* prlHashReads.c:164-310.  Rolling forward word + rolling reverse-complement word; canonical =
 * KmerSmaller(word, bal_word) ? word : bal_word; neighbour codes 0..3 or 4 = none.  N_kmer off. */
int sdto_chop_read(const uint8_t *src, int len, int K, int nw,
                   sdto_kmer *keys, uint8_t *prevc, uint8_t *nextc, uint64_t *hash)
{
	if (len < K + 1)          /* prlHashReads.c:592 */
		return 0;
	sdto_kmer word = k_zero(), bal;
	for (int i = 0; i < K; i++) {            /* :179-197 */
		word = k_shl2(word);
		word.w[3] |= src[i];
	}
	bal = sdto_reverse_complement(word, K);   /* :201 */
	int idx = 0;
	/* bal_seq[t] = int_comp(src[len-1-t]) (seq.c:102-120) */
#define BAL(t) ((uint8_t)(src[len - 1 - (t)] ^ 2))
	int bal_j = len - 1 - K;
	if (sdto_kmer_smaller(word, bal)) {        /* :215-222 */
		keys[idx] = word; prevc[idx] = 4; nextc[idx] = src[K];
	} else {                                   /* :223-230 */
		keys[idx] = bal; prevc[idx] = BAL(bal_j); nextc[idx] = 4;
	}
	if (hash) hash[idx] = sdto_hash_kmer(keys[idx], nw);
	idx++;
	for (int j = 1; j <= len - K; j++) {       /* :241-309 */
		word = sdto_next_kmer(word, src[j - 1 + K], K);
		bal_j = len - 1 - (j - 1 + K);
		bal = sdto_prev_kmer(bal, BAL(bal_j), K);
		if (sdto_kmer_smaller(word, bal)) {    /* :275-289 */
			keys[idx] = word;
			prevc[idx] = src[j - 1];
			nextc[idx] = (j < len - K) ? src[j + K] : 4;
		} else {                               /* :291-308 */
			keys[idx] = bal;
			prevc[idx] = (bal_j > 0) ? BAL(bal_j - 1) : 4;
			nextc[idx] = BAL(bal_j + K);
		}
		if (hash) hash[idx] = sdto_hash_kmer(keys[idx], nw);
		idx++;
	}
#undef BAL
	return idx;
}

/* ------------------------------------------------------------------ KmerSet */

#define FL_NULL(f, i)      (((f)[(i) >> 4] >> (((i) & 0x0f) << 1)) & 0x01)
#define FL_LIVE(f, i)      (!(((f)[(i) >> 4] >> (((i) & 0x0f) << 1)) & 0x03))
#define FL_SET_DEL(f, i)   ((f)[(i) >> 4] |= (0x02u << (((i) & 0x0f) << 1)))
#define FL_CLR_NULL(f, i)  ((f)[(i) >> 4] &= ~(0x01u << (((i) & 0x0f) << 1)))

/* newhash.c:116-141 is_prime_kh: trial division by odd i < (ubyte8)sqrt((float)num)  -- strict `<`
 * and a float-rounded argument, so squares of primes pass (survey 9.3-q7) */
static int is_prime_kh(uint64_t num)
{
	if (num < 4) return 1;
	if (num % 2 == 0) return 0;
	uint64_t max = (uint64_t)sqrt((float)num);
	for (uint64_t i = 3; i < max; i += 2)
		if (num % i == 0) return 0;
	return 1;
}

/* newhash.c:143-158 find_next_prime_kh */
uint64_t sdto_next_prime(uint64_t num)
{
	if (num % 2 == 0) num++;
	while (!is_prime_kh(num)) num += 2;
	return num;
}

/* newhash.c:160-193 init_kmerset.  NB max is computed with the *float* load factor before the
 * clamp, later growth uses the double copy (:176 vs :350). */
sdto_set *sdto_set_new(uint64_t init_size, float load_factor)
{
	sdto_set *s = (sdto_set *)malloc(sizeof *s);
	init_size = init_size < 3 ? 3 : sdto_next_prime(init_size);
	s->size = init_size;
	s->count = 0;
	s->max = s->size * load_factor;
	if (load_factor <= 0) load_factor = 0.25f;
	else if (load_factor >= 1) load_factor = 0.75f;
	s->load_factor = load_factor;
	s->array = (sdto_node *)calloc(s->size, sizeof(sdto_node));
	size_t fl = (s->size + 15) / 16 * 4;
	s->flags = (uint32_t *)malloc(fl);
	memset(s->flags, 0x55, fl);
	return s;
}

void sdto_set_free(sdto_set *s)
{
	if (!s) return;
	free(s->array); free(s->flags); free(s);
}

/* first probe slot: MER31 seq % size (newhash.c:428); MER63 128-bit % size (:423-425);
 * MER127 chunked 256-bit modulus (:43-55) */
uint64_t sdto_set_first_probe(const sdto_set *s, sdto_kmer seq, int nw)
{
	uint64_t size = s->size;
	if (nw == 1)
		return seq.w[3] % size;
	if (nw == 2) {
		unsigned __int128 t = ((unsigned __int128)seq.w[2] << 64) | seq.w[3];
		return (uint64_t)(t % size);
	}
	uint64_t t;
	t = (seq.w[0] % size) << 32 | (seq.w[1] >> 32 & 0xffffffff);
	t = (t % size) << 32 | (seq.w[1] & 0xffffffff);
	t = (t % size) << 32 | (seq.w[2] >> 32 & 0xffffffff);
	t = (t % size) << 32 | (seq.w[2] & 0xffffffff);
	t = (t % size) << 32 | (seq.w[3] >> 32 & 0xffffffff);
	t = (t % size) << 32 | (seq.w[3] & 0xffffffff);
	return t % size;
}

/* newhash.c:293-409 encap_kmerset: grow to a "prime" size and rehash IN PLACE, using the old flag
 * array as "not yet moved" marks and carrying evicted entries along (survey 9.6) */
static void set_encap(sdto_set *s, uint64_t num, int nw)
{
	if (s->count + num <= s->max)
		return;
	uint64_t n = s->size;
	do {
		if (n < 0xFFFFFFFU) n <<= 1;
		else n += 0xFFFFFFU;
		n = sdto_next_prime(n);
	} while (n * s->load_factor < s->count + num);
	s->array = (sdto_node *)realloc(s->array, n * sizeof(sdto_node));
	if (!s->array) { fprintf(stderr, "-- Out of memory --\n"); abort(); }
	size_t fl = (n + 15) / 16 * 4;
	uint32_t *newflags = (uint32_t *)malloc(fl);
	memset(newflags, 0x55, fl);
	uint64_t oldsize = s->size;
	uint32_t *oldflags = s->flags;
	s->size = n;
	s->max = n * s->load_factor;
	s->flags = newflags;
	for (uint64_t i = 0; i < oldsize; i++) {
		if (!FL_LIVE(oldflags, i))
			continue;
		sdto_node key = s->array[i];
		FL_SET_DEL(oldflags, i);
		for (;;) {
			uint64_t hc = sdto_set_first_probe(s, key.seq, nw);
			while (!FL_NULL(s->flags, hc)) {
				hc++;
				if (hc == s->size) hc = 0;
			}
			FL_CLR_NULL(s->flags, hc);
			if (hc < oldsize && FL_LIVE(oldflags, hc)) {
				sdto_node tmp = key;
				key = s->array[hc];
				s->array[hc] = tmp;
				FL_SET_DEL(oldflags, hc);
			} else {
				s->array[hc] = key;
				break;
			}
		}
	}
	free(oldflags);
}

/* newhash.c:411-462 put_kmerset (+ set_new_kmer :98-114, update_kmer :71-96) */
int sdto_set_put(sdto_set *s, sdto_kmer seq, int left, int right, int nw, uint64_t *slot)
{
	set_encap(s, 1, nw);
	uint64_t hc = sdto_set_first_probe(s, seq, nw);
	for (;;) {
		if (FL_NULL(s->flags, hc)) {
			FL_CLR_NULL(s->flags, hc);
			sdto_node *m = s->array + hc;
			memset(m, 0, sizeof *m);
			m->seq = seq;
			m->single = 1;                      /* empty_kmer has single = 1 (newhash.c:32-40) */
			m->count = 1;
			if (left < 4)  m->l_links |= 1u << (left * 6);
			if (right < 4) m->r_links |= 1u << (right * 6);
			s->count++;
			if (slot) *slot = hc;
			return 0;
		}
		if (sdto_kmer_equal(s->array[hc].seq, seq)) {
			sdto_node *m = s->array + hc;
			m->count++;
			if (left < 4) {
				uint32_t cov = (m->l_links >> (left * 6)) & 0x3f;
				if (cov < 63) m->l_links += 1u << (left * 6);
			}
			if (right < 4) {
				uint32_t cov = (m->r_links >> (right * 6)) & 0x3f;
				if (cov < 63) m->r_links += 1u << (right * 6);
			}
			m->single = 0;
			if (slot) *slot = hc;
			return 1;
		}
		hc++;
		if (hc == s->size) hc = 0;
	}
}

/* newhash.c:239-283 search_kmerset */
int sdto_set_search(const sdto_set *s, sdto_kmer seq, int nw, uint64_t *slot)
{
	uint64_t hc = sdto_set_first_probe(s, seq, nw);
	for (;;) {
		if (FL_NULL(s->flags, hc))
			return 0;
		if (sdto_kmer_equal(s->array[hc].seq, seq)) {
			if (slot) *slot = hc;
			return 1;
		}
		hc++;
		if (hc == s->size) hc = 0;
	}
}

/* ------------------------------------------------------------------ pass-1 driver */

/* prlHashReads.c:402-423: thrd_num sets, init_kmerset(1024, 0.77f).  -a n (initKmerSetSize, pregraph.c:160-162)
 * with n != 0: the MER63 / MER127 builds ask for k * 0xFFFFFF slots with k == 0 (:404-413), which init_kmerset
 * turns into 3 (newhash.c:163-166); the MER31 build ignores -a (:414-416). */
sdto_sets *sdto_sets_new_a(int nsets, int nw, int K, int init_kmerset_size)
{
	sdto_sets *S = (sdto_sets *)calloc(1, sizeof *S);
	S->nsets = nsets; S->nw = nw; S->K = K;
	S->sets = (sdto_set **)calloc(nsets, sizeof(sdto_set *));
	for (int i = 0; i < nsets; i++)
		S->sets[i] = sdto_set_new(init_kmerset_size && nw > 1 ? 0 : 1024, 0.77f);
	return S;
}

sdto_sets *sdto_sets_new(int nsets, int nw, int K)
{
	return sdto_sets_new_a(nsets, nw, K, 0);
}

void sdto_sets_free(sdto_sets *S)
{
	if (!S) return;
	for (int i = 0; i < S->nsets; i++)
		sdto_set_free(S->sets[i]);
	free(S->sets);
	free(S->patch);
	free(S);
}

/* signals 2 then 1 for one read: chopKmer4read, then every record goes to set hash % thrd_num
 * (prlHashReads.c:77-104,126-130).  Insertion order inside a set = read order, then position order,
 * exactly as the reference's batch scan. */
void sdto_sets_add_read(sdto_sets *S, const uint8_t *codes, int len)
{
	enum { MAXL = 8192 };
	static sdto_kmer keys[MAXL];
	static uint8_t pc[MAXL], nc[MAXL];
	static uint64_t hb[MAXL];
	if (len > MAXL) len = MAXL;
	int n = sdto_chop_read(codes, len, S->K, S->nw, keys, pc, nc, hb);     /* 0 records for a short read; it still has an ordinal */
	S->kmers_in_reads += (uint64_t)n;
	for (int i = 0; i < n; i++) {
		sdto_set *set = S->sets[hb[i] % (uint64_t)S->nsets];
		uint64_t slot;
		if (!sdto_set_put(set, keys[i], pc[i], nc[i], S->nw, &slot))
			set->array[slot].first = (S->reads_seen << 16) | (uint64_t)i;
	}
	S->reads_seen++;
}

void sdto_sets_add_reads(sdto_sets *S, const uint8_t *codes, const uint64_t *offsets, uint64_t nreads)
{
	for (uint64_t r = 0; r < nreads; r++)
		sdto_sets_add_read(S, codes + offsets[r], (int)(offsets[r + 1] - offsets[r]));
}

uint64_t sdto_sets_node_count(const sdto_sets *S)
{
	uint64_t n = 0;
	for (int i = 0; i < S->nsets; i++)
		n += S->sets[i]->count;
	return n;
}

/* prlHashReads.c:844-887 thread_delow: links with 0 < v <= d are zeroed; a node left with no links
 * at all is marked deleted (nodes already without links are marked too) */
uint64_t sdto_sets_delow(sdto_sets *S, int d)
{
	uint64_t removed = 0;
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			sdto_node *rs = s->array + i;
			for (int b = 0; b < 4; b++) {
				uint32_t c = (rs->l_links >> (6 * b)) & 0x3f;
				if (c > 0 && c <= (uint32_t)d) rs->l_links &= ~(0x3fu << (6 * b));
				c = (rs->r_links >> (6 * b)) & 0x3f;
				if (c > 0 && c <= (uint32_t)d) rs->r_links &= ~(0x3fu << (6 * b));
			}
			if (rs->l_links == 0 && rs->r_links == 0) {
				rs->deleted = 1;
				removed++;
			}
		}
	}
	return removed;
}

/* prlHashReads.c:911-967 thread_mark: degree, linear flag, histogram bin = single ? 1 : max(sum of
 * left links, sum of right links); no `deleted` test (survey 9.3-q13) */
uint64_t sdto_sets_mark(sdto_sets *S, int64_t hist[257])
{
	uint64_t linear = 0;
	memset(hist, 0, 257 * sizeof(int64_t));
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			sdto_node *rs = s->array + i;
			int in_num = 0, out_num = 0, l_cvg = 0, r_cvg = 0;
			for (int b = 0; b < 4; b++) {
				int c = (rs->l_links >> (6 * b)) & 0x3f;
				if (c > 0) { in_num++; l_cvg += c; }
				c = (rs->r_links >> (6 * b)) & 0x3f;
				if (c > 0) { out_num++; r_cvg += c; }
			}
			if (rs->single) hist[1]++;
			else hist[l_cvg > r_cvg ? l_cvg : r_cvg]++;
			if (in_num == 1 && out_num == 1) {
				rs->linear = 1;
				linear++;
			}
		}
	}
	return linear;
}

/* prlHashReads.c:994-1023 freqStat */
int sdto_write_kmerfreq(const char *path, const int64_t hist[257])
{
	FILE *fo = fopen(path, "w");
	if (!fo) return -1;
	for (int i = 1; i < 256; i++)
		fprintf(fo, "%lld\n", (long long)hist[i]);
	fclose(fo);
	return 0;
}

uint64_t sdto_sets_export_first(const sdto_sets *S, uint64_t *first)
{
	uint64_t n = 0;
	for (int t = 0; t < S->nsets; t++) {
		const sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++)
			if (!FL_NULL(s->flags, i)) first[n++] = s->array[i].first;
	}
	return n;
}

uint64_t sdto_sets_export(const sdto_sets *S, uint64_t *keys4, uint32_t *l_links, uint32_t *r_links,
                          uint32_t *count, uint8_t *flags)
{
	uint64_t n = 0;
	for (int t = 0; t < S->nsets; t++) {
		const sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			const sdto_node *rs = s->array + i;
			if (keys4) memcpy(keys4 + 4 * n, rs->seq.w, 32);
			if (l_links) l_links[n] = rs->l_links;
			if (r_links) r_links[n] = rs->r_links;
			if (count) count[n] = rs->count;
			if (flags) flags[n] = (uint8_t)(rs->linear | (rs->deleted << 1) | (rs->single << 2));
			n++;
		}
	}
	return n;
}
