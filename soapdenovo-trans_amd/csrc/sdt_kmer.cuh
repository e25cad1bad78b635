// sdt_kmer.cuh -- device-side k-mer arithmetic for gfx950 (wave64).
//
// Device restatement of the reference's Kmer primitives (paths relative to /root/reference/src):
//   Kmer layout / base coding        inc/def.h:39-59
//   createFilter                     kmer.c:313-355
//   reverseComplement                kmer.c:548-656
//   KmerSmaller                      kmer.c:27-68
//   chopKmer4read record semantics   prlHashReads.c:164-310
// but NOT its structure: the reference rolls a forward and a reverse-complement word along each read
// (one thread per read); here every lane extracts "its" k-mer straight out of the packed 2-bit stream
// with a funnel shift, so 64 lanes cover 64 consecutive start positions of the LDS-resident tile and
// there is no serial dependence along the read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdt {

// Key<NW>: w[0] is the MOST significant word (reference struct order high1,low1,high2,low2).
template <int NW> struct Key { uint64_t w[NW]; };

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
	x ^= x >> 33;
	return x;
}

// slot / owner hash of a canonical key (our own choice: the reference's table position `seq % prime`
// and its owner hash_kmer % thrd_num are layout details, not results)
template <int NW> __host__ __device__ inline uint64_t key_hash(const Key<NW> &k)
{
	uint64_t h = 0x9E3779B97F4A7C15ULL;
#pragma unroll
	for (int i = 0; i < NW; i++)
		h = mix64(h ^ k.w[i]) + 0x632BE59BD9B4E019ULL * (uint64_t)(i + 1);
	return mix64(h);
}

template <int NW> __device__ inline bool key_less(const Key<NW> &a, const Key<NW> &b)
{
#pragma unroll
	for (int i = 0; i < NW; i++)
		if (a.w[i] != b.w[i])
			return a.w[i] < b.w[i];
	return false;
}

template <int NW> __device__ inline bool key_eq(const Key<NW> &a, const Key<NW> &b)
{
	bool e = true;
#pragma unroll
	for (int i = 0; i < NW; i++)
		e = e && (a.w[i] == b.w[i]);
	return e;
}

// reverse the order of the 32 two-bit groups of a word: full bit reversal, then swap the two bits
// inside each group back
__device__ inline uint64_t rev2bit(uint64_t x)
{
	x = __brevll(x);
	return ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
}

// big-integer right shift of NW words (w[0] most significant) by 0 < s < 64*NW bits
template <int NW> __device__ inline Key<NW> key_shr(const Key<NW> &k, int s)
{
	Key<NW> r;
	const int ws = s >> 6, bs = s & 63;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int src = i - ws;
		uint64_t v = 0;
#pragma unroll
		for (int j = 0; j < NW; j++) {      // select without dynamic register indexing
			if (j == src)
				v |= k.w[j] >> bs;
			if (j == src - 1 && bs)
				v |= k.w[j] << (64 - bs);
		}
		r.w[i] = v;
	}
	return r;
}

// reverse complement of a right-aligned K-mer: complement every base (x ^ 2 -> ^0xAAAA...), reverse
// the 2-bit groups across the whole register, right-align the 2K significant bits
template <int NW> __device__ inline Key<NW> key_revcomp(const Key<NW> &k, int K)
{
	Key<NW> r;
#pragma unroll
	for (int i = 0; i < NW; i++)
		r.w[i] = rev2bit(k.w[NW - 1 - i] ^ 0xAAAAAAAAAAAAAAAAULL);
	if (NW == 1) {
		r.w[0] >>= (64 - 2 * K);
		return r;
	}
	return key_shr<NW>(r, 64 * NW - 2 * K);
}

// Packed stream in LDS: 32-bit words, 16 bases each, first base in bits 31..30.  `lds` points at the
// word holding stream bit 0 of the tile and is preceded by LDS_LEAD readable words (their content is
// masked off), so windows that start "before" the tile need no branch.
constexpr int LDS_LEAD = 8;
constexpr int TAIL_PAD = 4;        // readable words past the last base of a tile

__device__ inline uint32_t stream_base(const uint32_t *lds, int p)
{
	return (lds[p >> 4] >> (30 - 2 * (p & 15))) & 3u;
}

// the K bases starting at base index p (p >= 0) as a right-aligned big-endian 2-bit integer
template <int NW> __device__ inline Key<NW> stream_kmer(const uint32_t *lds, int p, int K)
{
	Key<NW> k;
	// window of 64*NW bits that ENDS at the k-mer's last bit
	const int s = 2 * (p + K) - 64 * NW;            // may be negative: covered by LDS_LEAD
	const int wi = s >> 5;                          // arithmetic shift: floor
	const int sh = s & 31;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const uint64_t a = lds[wi + 2 * i], b = lds[wi + 2 * i + 1], c = lds[wi + 2 * i + 2];
		const uint64_t hi = (a << 32) | b;
		k.w[i] = (hi << sh) | (c >> (32 - sh));     // (c holds 32 bits in a 64-bit word: sh == 0 shifts all of them out, no branch)
	}
	// createFilter (kmer.c:313-355): keep the low 2K bits.  With 4-word keys K may be as small as 65, so
	// more than one leading word can be (partly) outside the k-mer.
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);     // significant bits in word i
		if (bits <= 0)
			k.w[i] = 0;
		else if (bits < 64)
			k.w[i] &= (1ULL << bits) - 1ULL;
	}
	return k;
}

// One chopKmer4read record (prlHashReads.c:215-230,275-308; survey 9.1): j = k-mer index inside a
// read of `len` bases that starts at stream base `rb`.  prev/next: 0..3 or 4 = none.
template <int NW>
__device__ inline Key<NW> chop_record(const uint32_t *lds, int rb, int len, int j, int K, uint32_t &prev, uint32_t &next)
{
	const int p = rb + j;
	const Key<NW> fw = stream_kmer<NW>(lds, p, K);
	const Key<NW> rc = key_revcomp<NW>(fw, K);
	const bool has_l = j > 0, has_r = j < len - K;
	// (both neighbours are read whether they exist or not: the lead / tail words of the tile make that legal, and two
	// conditional LDS reads were two exec-mask branches per k-mer)
	const uint32_t lb = stream_base(lds, p - 1);
	const uint32_t rbse = stream_base(lds, p + K);
	if (key_less<NW>(fw, rc)) {
		prev = has_l ? lb : 4u;
		next = has_r ? rbse : 4u;
		return fw;
	}
	prev = has_r ? (rbse ^ 2u) : 4u;
	next = has_l ? (lb ^ 2u) : 4u;
	return rc;
}

// append base b at the low end (the caller masks the result down to K bases where it matters)
template <int NW> __device__ inline Key<NW> key_append(const Key<NW> &k, uint32_t b)
{
	Key<NW> r;
#pragma unroll
	for (int i = 0; i < NW; i++)
		r.w[i] = (k.w[i] << 2) | (i + 1 < NW ? k.w[i + 1] >> 62 : (uint64_t)b);
	return r;
}

} // namespace sdt
