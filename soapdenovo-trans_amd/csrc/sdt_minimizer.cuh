// sdt_minimizer.cuh -- the minimizer bucket of a k-mer: what the locality pipeline files k-mers by and what decides the rank that
// owns a k-mer (sdt_kmer_owner).
//
// A k-mer's bucket is a function of its canonical minimizer: the smallest hash among the canonical m-mers it contains
// (m = 7..11, w = K - m + 1 of them).  A k-mer and its reverse complement contain the same canonical m-mers, so every
// occurrence of a canonical key -- on either strand, in any read -- has the same bucket (sdt_superkmer.cuh).
#pragma once
#include "sdt_kmer.cuh"

namespace sdt {

#ifndef SDT_SK_L1BITS
#define SDT_SK_L1BITS 8
#endif
constexpr int SK_L1BITS = SDT_SK_L1BITS;
#ifndef SDT_SK_L2BITS
#define SDT_SK_L2BITS 10
#endif
constexpr int SK_L2BITS = SDT_SK_L2BITS;
constexpr int SK_NB1 = 1 << SK_L1BITS;
constexpr int SK_NB2 = 1 << SK_L2BITS;
constexpr int SK_NBF = SK_NB1 * SK_NB2;          // final buckets

// minimizer length for a k-mer size (window w = K - m + 1 m-mers)
__host__ __device__ inline int sk_minimizer_len(int K) { return K >= 23 ? 11 : (K >= 17 ? 9 : 7); }

// order of the canonical m-mers (which m-mer wins is a layout detail; one 32-bit multiply: v_mul_lo_u32 is quarter rate)
__host__ __device__ inline uint32_t sk_mmer_hash(uint32_t canon)
{
	uint32_t h = (canon + 0x7F4A7C15u) * 0x9E3779B1u;
	return h ^ (h >> 15);
}

// bucket hash of a k-mer = a second mix of its smallest m-mer hash (the minimum itself is biased towards 0); the
// buckets are its TOP bits
__host__ __device__ inline uint32_t sk_bucket_hash(uint32_t hvmin)
{
	uint32_t h = (hvmin ^ 0x5BD1E995u) * 0x85EBCA77u;
	return h ^ (h >> 13);
}
__host__ __device__ inline uint32_t sk_final_bucket(uint32_t bh) { return bh >> (32 - SK_L1BITS - SK_L2BITS); }
__host__ __device__ inline uint32_t sk_l1_bucket(uint32_t bh) { return bh >> (32 - SK_L1BITS); }
__host__ __device__ inline uint32_t sk_l2_bucket(uint32_t bh) { return (bh >> (32 - SK_L1BITS - SK_L2BITS)) & (SK_NB2 - 1); }

// reverse the order of the 16 two-bit groups of a 32-bit word
__host__ __device__ inline uint32_t sk_rev2bit32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	x = __brev(x);
#else
	x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
	x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
	x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
	x = (x >> 16) | (x << 16);
#endif
	return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}

// canonical m-mer (m <= 15) from its right-aligned forward value
__host__ __device__ inline uint32_t sk_canon_mmer(uint32_t fw, int m)
{
	const uint32_t rc = sk_rev2bit32(fw ^ 0xAAAAAAAAu) >> (32 - 2 * m);
	return fw < rc ? fw : rc;
}

// the low 32 bits of (key >> s), key = NW words, w[0] most significant, 0 <= s < 64 * NW
template <int NW> __host__ __device__ inline uint32_t key_bits32(const Key<NW> &k, int s)
{
	if (NW == 1)
		return (uint32_t)(k.w[0] >> s);
	const int ws = s >> 6, bs = s & 63;
	uint64_t lo = 0, hi = 0;
#pragma unroll
	for (int j = 0; j < NW; j++) {                       // (select without dynamic register indexing)
		if (j == NW - 1 - ws) lo = k.w[j];
		if (j == NW - 2 - ws) hi = k.w[j];
	}
	return (uint32_t)((lo >> bs) | (bs ? hi << (64 - bs) : 0ULL));
}

// smallest m-mer hash of a k-mer held as a right-aligned key (either strand: the canonical m-mers are the same)
template <int NW> __host__ __device__ inline uint32_t key_min_mmer_hash(const Key<NW> &k, int K)
{
	const int m = sk_minimizer_len(K);
	const uint32_t mm = (1u << (2 * m)) - 1u;
	uint32_t best = 0xFFFFFFFFu;
	for (int s = 2 * (K - m); s >= 0; s -= 2) {          // the m-mer that starts at base (K - m) - s / 2 of the k-mer
		const uint32_t hv = sk_mmer_hash(sk_canon_mmer(key_bits32<NW>(k, s) & mm, m));
		best = hv < best ? hv : best;
	}
	return best;
}

// final bucket (0 .. SK_NBF - 1) of a k-mer: what k_sk_scatter_reads(_seq) + k_sk_scatter_records file its occurrences under
template <int NW> __host__ __device__ inline uint32_t key_final_bucket(const Key<NW> &k, int K)
{
	return sk_final_bucket(sk_bucket_hash(key_min_mmer_hash<NW>(k, K)));
}

} // namespace sdt
