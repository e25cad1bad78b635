// sdt_tile.cuh -- a tile of packed reads staged in LDS: what k_count_reads (sdt_table_kernels.cuh) and the strip form of the level-1
// scatter (sdt_superkmer_kernels.cuh) cut their k-mers out of.  Included after sdt_kmer.cuh (chop_record lives there).
#pragma once
constexpr int TILE_READS = 64;     // reads staged per workgroup tile

// Stage reads [r0, r1) of the batch in LDS.  Returns the tile's k-mer count; fills
//   s_words : LDS_LEAD lead words, then the packed words that hold bases [off[r0], off[r1])
//   s_rb[i] : stream base index of read r0+i relative to the first staged word (i = 0..nr)
//   s_pre[i]: exclusive prefix sum of k-mers per read (i = 0..nr)
struct TileView {
	const uint32_t *words;  // points at the first staged word (after the lead)
	const uint32_t *rb;
	const uint32_t *pre;
	int nr;
	uint32_t nk;
};

__device__ inline TileView stage_tile(uint32_t *smem, int max_tile_words, const uint32_t *__restrict__ packed,
                                      const uint64_t *__restrict__ offs, uint64_t r0, uint64_t nreads, int K,
                                      int tile_reads = TILE_READS)
{
	uint32_t *s_rb = smem;                           // TILE_READS + 1
	uint32_t *s_pre = smem + (TILE_READS + 1);       // TILE_READS + 1
	uint32_t *s_words = smem + 2 * (TILE_READS + 1) + 2;   // keep 16-byte alignment irrelevant: b32 reads
	const int tid = threadIdx.x;
	const int nr = (int)((nreads - r0) < (uint64_t)tile_reads ? (nreads - r0) : (uint64_t)tile_reads);
	const uint64_t base0 = offs[r0];
	const uint64_t word0 = base0 >> 4;
	const uint64_t base_end = offs[r0 + nr];
	const uint64_t word_end = (base_end + 15) >> 4;
	int nwords = (int)(word_end - word0) + TAIL_PAD;
	if (nwords > max_tile_words)
		nwords = max_tile_words;                     // cannot happen when max_read_len was honoured
	// per-read geometry
	if (tid <= nr) {
		const uint64_t o = offs[r0 + tid];
		s_rb[tid] = (uint32_t)(o - (word0 << 4));
		uint32_t nk = 0;
		if (tid < nr) {
			const uint64_t len = offs[r0 + tid + 1] - o;
			nk = len >= (uint64_t)(K + 1) ? (uint32_t)(len - K + 1) : 0u;    // prlHashReads.c:592
		}
		s_pre[tid] = nk;
	}
	// coalesced copy of the packed words (zero lead: its content is masked off anyway)
	if (tid < LDS_LEAD)
		s_words[tid] = 0;
	for (int i = tid; i < nwords; i += TPB)
		s_words[LDS_LEAD + i] = packed[word0 + i];
	__syncthreads();
	// exclusive scan of <= 65 values by one wave (two values per lane)
	if (tid < 64) {
		uint32_t a = tid < nr ? s_pre[tid] : 0u;
		uint32_t x = a;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t y = __shfl_up(x, d);
			if (tid >= d)
				x += y;
		}
		s_pre[tid] = x - a;
		if (tid == 63)
			s_pre[64] = x;
	}
	__syncthreads();
	TileView tv;
	tv.words = s_words + LDS_LEAD;
	tv.rb = s_rb;
	tv.pre = s_pre;
	tv.nr = nr;
	tv.nk = s_pre[64];
	return tv;
}

// find the read that owns k-mer q of the tile: largest i with pre[i] <= q (reads with 0 k-mers are skipped
// automatically because their interval is empty)
__device__ inline int tile_find_read(const uint32_t *pre, uint32_t q)
{
	int lo = 0, hi = TILE_READS;                     // pre[64] = total > q
#pragma unroll
	for (int s = 0; s < 6; s++) {
		const int mid = (lo + hi) >> 1;
		if (pre[mid] <= q) lo = mid; else hi = mid;
	}
	return lo;
}

static_assert(TILE_READS == 64, "the scan and the binary search assume 64 reads per tile");
