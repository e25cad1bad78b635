// sdt_partition.cuh -- the locality pipeline for 1-word keys (K <= 31) on gfx950.
//
// Why: memory-side 64-bit atomics run at ~19-24 G/s on MI355X whatever the footprint, scope or lane
// locality (profiles/r1/microbench_atomics.txt), and the direct kernel (k_count_reads: one atomic per k-mer
// occurrence) sits on that ceiling.  Transcriptome reads repeat the same k-mers hundreds of times, so the
// way past the ceiling is to bring the occurrences of a key TOGETHER and count them in LDS, where
// atomics are cheap, and touch the global node table once per distinct key per batch.
//
// Pipeline per batch of reads (all streaming, 8-byte records):
//   k_part_hist   chop every read (same chop_record as the direct path), h = bij(key) (a bijection on 2K bits),
//                 histogram of the top L1BITS+L2BITS bits of h in LDS  -> exact bucket sizes
//   k_part_scan   exclusive scan -> offsets of the 2^(L1+L2) final buckets, cursors, tile tables
//   k_part_l1     chop again, scatter records (h low bits << 5 | neighbour code) into L1 buckets of buffer A
//   k_part_l2     per L1 bucket: scatter into its 2^L2BITS sub-buckets in buffer B (final buckets contiguous)
//   k_part_final  one workgroup per final bucket: LDS hash table keyed by the remaining bits of h with the
//                 same [count:16 | r_links | l_links] word as the global table, updated by LDS CAS; at the end
//                 every LDS entry is merged into the global table with ONE saturating CAS (node_merge).
//                 Records that do not fit (LDS table full, 16-bit count about to wrap) go to the global
//                 table directly (table_put), so correctness never depends on the bucket count.
// The reference's semantics are untouched: node state = f(multiset of (key, prev, next)) (survey 9.1) and
// min(63, a + b) merges are exact because the per-field counters saturate at the same bound.
#pragma once
#include "sdt_kmer.cuh"
#include "sdt_table.cuh"

namespace sdt {

constexpr int L1BITS = 7;
constexpr int L2BITS = 7;
constexpr int NB1 = 1 << L1BITS;
constexpr int NB2 = 1 << L2BITS;
constexpr int NBF = NB1 * NB2;            // final buckets
constexpr int PT_TPB = 256;               // partition kernels
constexpr int L2_TILE = 8192;             // records per k_part_l2 tile
constexpr int FIN_TPB = 512;
constexpr int FIN_SLOTS = 4096;           // LDS table entries of k_part_final (2 x 32 KiB)
constexpr int FIN_MAX_FILL = FIN_SLOTS * 7 / 8;

// ---- bijection on n = 2K bits (xorshift / odd multiply are invertible mod 2^n) -----------------------
constexpr uint64_t BIJ_C1 = 0xff51afd7ed558ccdULL, BIJ_C2 = 0xc4ceb9fe1a85ec53ULL;

constexpr uint64_t mod_inverse(uint64_t a)
{
	uint64_t x = a;                       // Newton: 3 correct bits -> 64
	for (int i = 0; i < 6; i++)
		x *= 2 - a * x;
	return x;
}
constexpr uint64_t BIJ_I1 = mod_inverse(BIJ_C1), BIJ_I2 = mod_inverse(BIJ_C2);
static_assert(BIJ_C1 * BIJ_I1 == 1 && BIJ_C2 * BIJ_I2 == 1, "modular inverses");

__host__ __device__ inline uint64_t bij_fwd(uint64_t x, int n)
{
	const uint64_t m = n >= 64 ? ~0ULL : ((1ULL << n) - 1);
	const int s = n / 2;
	x ^= x >> s; x = (x * BIJ_C1) & m;
	x ^= x >> s; x = (x * BIJ_C2) & m;
	x ^= x >> s;
	return x;
}
// inverse of x ^= x >> s for s >= n/2 (n even or odd: s = n/2 gives 2s >= n-1): one more xor undoes it
__host__ __device__ inline uint64_t unxorshift(uint64_t y, int s, int n)
{
	uint64_t x = y;
	for (int i = s; i < n; i += s)
		x = y ^ (x >> s);
	return x;
}
__host__ __device__ inline uint64_t bij_inv(uint64_t x, int n)
{
	const uint64_t m = n >= 64 ? ~0ULL : ((1ULL << n) - 1);
	const int s = n / 2;
	x = unxorshift(x, s, n);
	x = (x * BIJ_I2) & m;
	x = unxorshift(x, s, n);
	x = (x * BIJ_I1) & m;
	x = unxorshift(x, s, n);
	return x;
}

// geometry of the hashed key for a given K
struct PartGeom {
	int n;            // 2K bits
	int tagbits;      // n - L1BITS - L2BITS : bits kept in the final record / LDS tag
};

struct PartBufs {
	unsigned int *hist;               // NBF        : records per final bucket (this batch)
	unsigned long long *off2;         // NBF + 1    : exclusive scan of hist
	unsigned long long *cursor1;      // NB1        : write cursors into A
	unsigned long long *cursor2;      // NBF        : write cursors into B
	unsigned int *tile1;              // NB1 + 1    : first k_part_l2 tile of each L1 bucket
	uint64_t *A, *B;                  // record buffers
};

__device__ inline uint64_t make_record(uint64_t h, uint32_t prev, uint32_t next, int keepbits)
{
	return ((h & ((1ULL << keepbits) - 1)) << 5) | (uint64_t)(prev * 5u + next);
}

} // namespace sdt
