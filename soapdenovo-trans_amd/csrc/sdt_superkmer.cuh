// sdt_superkmer.cuh -- the locality pipeline of pass 1 on gfx950: minimizer buckets of super-k-mers.
//
// Why.  put_kmerset (newhash.c:411-462) is one random read-modify-write of a node per k-mer occurrence, and on
// MI355X a memory-side atomic costs the same ~20 G/s whatever the footprint (profiles/r1/microbench_atomics.txt):
// the direct kernel (k_count_reads) sits on that ceiling.  Transcriptome reads repeat a k-mer hundreds of
// times, so the way past it is to bring the occurrences of a key TOGETHER, count them where atomics are cheap
// (LDS), and touch the node table once per distinct key.
//
// How.  A k-mer's bucket is a function of its canonical minimizer: the smallest hash among the canonical
// m-mers it contains (m = 7..11).  A k-mer and its reverse complement contain the same canonical m-mers, so
// every occurrence of a canonical key -- on either strand, in any read -- falls into the same bucket.
// Consecutive k-mers of a read mostly share their minimizer; a maximal run with one bucket is cut out of the
// read as ONE record (a "super-k-mer"): 8 B of header + the run's bases with one base of context either side
// (the prev / next neighbour codes of chopKmer4read, prlHashReads.c:215-230,275-308, are read off them).  That
// is ~3 B per k-mer occurrence (24-byte records of ~8 k-mers at K = 31) instead of the 16-B (key, meta) record of the
// per-k-mer exchange.
//
//   k_sk_scatter_reads(_seq)  chop + minimizers per tile of reads in LDS (_seq: one lane walks one read, rolling m-mers and a
//                        block sliding minimum; otherwise strips of 64 positions per wave), cut the runs, append each record to
//                        its level-1 bucket (256 of them).  Space comes from a pool of fixed-size chunks; every
//                        workgroup owns one open chunk per bucket and reserves slots with LDS atomics, so the
//                        only global atomic is the pool bump once per 128 chunks.
//   k_sk_scatter_records level 2: every level-1 bucket is split 1024 ways the same way (records only move).
//   k_sk_count           one workgroup per final bucket (2^18): records -> LDS -> k-mers -> an LDS hash table
//                        whose entries have the layout of the node table's (key, val); LDS atomics do the
//                        counting (wave64, 4096 slots); at the end every LDS entry is merged into the node table
//                        with ONE saturating CAS (table_merge).  min(63, a + b) per 6-bit link counter and the
//                        plain sum of counts is exactly what replaying the occurrences one by one leaves
//                        (newhash.c:71-96), so the node table -- and *.kmerFreq -- is bit-identical.
// Nothing depends on a bucket fitting: a full LDS table is flushed and refilled, a k-mer that finds no LDS slot
// and a record that finds no chunk go through table_put directly.
#pragma once
#include "sdt_kmer.cuh"
#include "sdt_minimizer.cuh"
#include "sdt_table.cuh"

namespace sdt {

constexpr int SK_POSBITS = 22 - SK_L2BITS;          // header: level-2 bucket and position share 22 bits
constexpr int SK_CAP1 = 32;                      // records per level-1 chunk
#ifndef SDT_SK_CAP2
#define SDT_SK_CAP2 16
#endif
constexpr int SK_CAP2 = SDT_SK_CAP2;             // records per level-2 chunk (16 or 32: the fill of a chunk rides in the top bits of its list entry)
static_assert(SK_CAP2 == 16 || SK_CAP2 == 32, "a level-2 chunk holds 16 or 32 records");
constexpr uint32_t SK_NOCHUNK = 0xFFFFFFFFu;
constexpr int SK_MAX_RUN = 64;                   // k-mers per record (6-bit field holds n - 1)

// record: REC_WORDS 64-bit words
//   [0] read ordinal (34 bits) << 30 | position of the run's first k-mer in its read (12 bits) << 18 |
//       level-2 bucket (10 bits) << 8 | (n - 1) << 2 | has_prev << 1 | has_next
//   [1..] the bases [first k-mer - has_prev, last k-mer + K + has_next), 2 bits each, first base in the MOST
//         significant pair of word 1 (the packed-read convention of include/sdt_gpu.h), zero padded
// (the level-1 bucket of a record is where it lies; the level-2 bucket rides in the header: 24 / 40 / 56 bytes per record)
template <int NW> struct SkFmt {
	static constexpr int BW = NW == 1 ? 2 : (NW == 2 ? 4 : 6);     // base words: 64 / 128 / 192 bases
	static constexpr int REC_WORDS = 1 + BW;                       // 24 / 40 / 56 bytes
	static constexpr int CAP_BASES = 32 * BW;
	// The slots of a LEVEL-2 chunk.  The memory side works in aligned 64-byte blocks (tools/box_probe.hip `pieces`: scattered writes of
	// 96 bytes run at 1.43 TB/s, of 128 aligned bytes at 5.2), and the groups of four 24-byte records the level-2 scatter writes are 96
	// bytes.  SDT_SK_REC2_PAD=1 gives the records 32-byte slots (the fourth word written as zero: a group is one aligned 128-byte line)
	// -- measured: the level-2 scatter got SLOWER, 64.7 / 66.7 -> 69.4 / 68.6 ms per step (a third more bytes; the kernel is not bound by
	// the write rate but by its own phases: two barriers and a serial book-keeping section per tile of 512 records at 8 waves per CU).
	// Kept as a build knob, off.
#ifndef SDT_SK_REC2_PAD
#define SDT_SK_REC2_PAD 0
#endif
	static constexpr int REC2_STRIDE = (NW == 1 && SDT_SK_REC2_PAD) ? 4 : REC_WORDS;
};
constexpr uint64_t SK_MAX_READ_ORDINAL = 1ULL << 34;             // reads of one run the header can number
constexpr int SK_MAX_READ_LEN = (1 << SK_POSBITS) - 1;                            // positions the header can hold

__host__ __device__ inline int sk_rec_words(int nw) { return nw == 1 ? 3 : (nw == 2 ? 5 : 7); }
__host__ __device__ inline int sk_rec2_stride(int nw) { return nw == 1 ? SkFmt<1>::REC2_STRIDE : (nw == 2 ? SkFmt<2>::REC2_STRIDE : SkFmt<4>::REC2_STRIDE); }

// longest run a record can hold: n + K - 1 bases + 2 context bases must fit the base words; a power of two
__host__ __device__ inline int sk_max_run(int K, int nw)
{
	const int cap = 32 * (nw == 1 ? 2 : (nw == 2 ? 4 : 6)) - K - 1;
	int n = SK_MAX_RUN;
	while (n > cap)
		n >>= 1;
	return n;
}

// the m bases starting at base index p of a packed word stream (16 bases per uint32, first base in bits 31..30)
__host__ __device__ inline uint32_t sk_stream_mmer(const uint32_t *words, int p, int m)
{
	const int s = 2 * p, wi = s >> 5, sh = s & 31;
	const uint64_t win = ((uint64_t)words[wi] << 32) | words[wi + 1];
	return (uint32_t)((win << sh) >> (64 - 2 * m));
}

// 32 bases starting at base index p as one 64-bit word (first base most significant)
__host__ __device__ inline uint64_t sk_stream_word(const uint32_t *words, int p)
{
	const int s = 2 * p, wi = s >> 5, sh = s & 31;
	const uint64_t hi = ((uint64_t)words[wi] << 32) | words[wi + 1];
	return sh ? ((hi << sh) | ((uint64_t)words[wi + 2] >> (32 - sh))) : hi;
}

__host__ __device__ inline uint64_t sk_header(uint64_t read_ord, uint32_t pos, uint32_t l2, int n, int has_prev, int has_next)
{
	return (read_ord << 30) | ((uint64_t)(pos & (uint32_t)SK_MAX_READ_LEN) << (8 + SK_L2BITS)) | ((uint64_t)(l2 & (uint32_t)(SK_NB2 - 1)) << 8) | ((uint64_t)(n - 1) << 2) |
	       ((uint64_t)has_prev << 1) | (uint64_t)has_next;
}
__host__ __device__ inline int sk_hdr_n(uint64_t h) { return (int)((h >> 2) & 63u) + 1; }
__host__ __device__ inline int sk_hdr_prev(uint64_t h) { return (int)((h >> 1) & 1u); }
__host__ __device__ inline int sk_hdr_next(uint64_t h) { return (int)(h & 1u); }
__host__ __device__ inline uint32_t sk_hdr_l2(uint64_t h) { return (uint32_t)((h >> 8) & (uint32_t)(SK_NB2 - 1)); }
__host__ __device__ inline uint32_t sk_hdr_pos(uint64_t h) { return (uint32_t)((h >> (8 + SK_L2BITS)) & (uint32_t)SK_MAX_READ_LEN); }
__host__ __device__ inline uint64_t sk_hdr_read(uint64_t h) { return h >> 30; }
static_assert(SK_L2BITS >= 10 && SK_L2BITS <= 12, "the record header holds a 10..12-bit level-2 bucket");
constexpr uint32_t SK_HDR_KIND_MASK = (1u << (8 + SK_L2BITS)) - 1u;     // bucket, n, context flags: what identical records share besides their bases

// device state of the pipeline (all device pointers)
struct SkItem { uint32_t b1, c0, c1, pad; };      // level 2: one work item = a run [c0, c1) of the chunk list of ONE level-1 bucket
constexpr int SK_TILE_READS = 32;                // reads per tile of k_sk_scatter_reads (half of k_count_reads': LDS for 6 workgroups per CU)
struct SkPool {
	uint64_t *recs;            // chunks * cap * REC_WORDS words
	uint32_t *meta;            // per chunk: bucket (24 bits) | records in use << 24
	uint32_t *next;            // bump allocator: chunks handed out
	uint32_t chunks;           // capacity
};

} // namespace sdt
