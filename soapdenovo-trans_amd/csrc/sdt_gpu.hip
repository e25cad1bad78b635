// sdt_gpu.hip -- kernels + C ABI (include/sdt_gpu.h) of the MI355X-native pregraph hashing path.
//
// gfx950 only.  No CPU fallback: every entry point needs a live HIP device.
//
// Kernels (all integer / HBM-bound, no MFMA):
//   k_count_reads<NW>     chopKmer4read (prlHashReads.c:164-310) fused with put_kmerset
//                         (newhash.c:411-462): a workgroup stages a tile of packed reads in LDS with
//                         coalesced loads, every lane cuts its k-mers out of LDS by funnel shift and
//                         updates the node table with one 64-bit atomic per occurrence.
//   k_sk_*                the locality pipeline (sdt_superkmer.cuh): minimizer buckets of super-k-mer records counted in LDS
//   k_delow<NW>           thread_delow   (prlHashReads.c:844-887)
//   k_mark_hist<NW>       thread_mark    (prlHashReads.c:911-967) + per-thread kmerFreq bins
//   k_export<NW>          compaction of the table into kmer_t-shaped arrays (inc/newhash.h:65-77)
//   k_rehash<NW>          table growth (the analogue of encap_kmerset, newhash.c:293-409)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <new>
#include <thread>
#include <mutex>
#include <atomic>

#include "sdt_internal.hpp"
#include "sdt_superkmer.cuh"

using namespace sdt;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static std::atomic<int> g_live_ctx{0};                       // contexts alive in this process (the arena is trimmed when the last one goes)

int sdti::fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}
using sdti::fail;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
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

template <int NW>
__global__ __launch_bounds__(TPB) void k_count_reads(const uint32_t *__restrict__ packed,
                                                     const uint64_t *__restrict__ offs, uint64_t nreads, int K,
                                                     int max_tile_words, Table<NW> tbl, Stats *stats,
                                                     uint64_t ord_base, uint64_t ord_stride)
{
	extern __shared__ uint32_t smem[];
	const uint64_t ntiles = (nreads + TILE_READS - 1) / TILE_READS;
	uint32_t claimed = 0, failed = 0, done = 0;
	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const TileView tv = stage_tile(smem, max_tile_words, packed, offs, tile * TILE_READS, nreads, K);
		for (uint32_t q = threadIdx.x; q < tv.nk; q += TPB) {
			const int r = tile_find_read(tv.pre, q);
			const int j = (int)(q - tv.pre[r]);
			const int len = (int)(tv.rb[r + 1] - tv.rb[r]);
			uint32_t prev, next;
			const Key<NW> key = chop_record<NW>(tv.words, (int)tv.rb[r], len, j, K, prev, next);
			// ordinal of this occurrence in the reference's stream order: (read ordinal, position in read)
			const uint64_t ord = tbl.first ? ((ord_base + (tile * TILE_READS + (uint64_t)r) * ord_stride) << 16) | (uint64_t)j : ORD_NONE;
			if (!table_put<NW>(tbl, key, prev, next, claimed, ord))
				failed++;
			done++;
		}
		__syncthreads();                             // tile buffer is reused
	}
	// per-wave reduction of the counters, one atomic per wave
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		claimed += __shfl_down(claimed, d);
		failed += __shfl_down(failed, d);
		done += __shfl_down(done, d);
	}
	if ((threadIdx.x & 63) == 0) {
		if (done) atomicAdd(&stats->kmers, (unsigned long long)done);
		if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
		if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
	}
}

template <int NW> __global__ __launch_bounds__(TPB) void k_clear(Table<NW> tbl)
{
	const uint64_t slots = tbl.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		Entry<NW> e;
#pragma unroll
		for (int i = 0; i < NW; i++)
			e.key[i] = KEY_EMPTY;
		e.val = 0;
		tbl.ent[s] = e;
		tbl.aux[s] = 0;
		if (tbl.first)
			tbl.first[s] = ORD_NONE;
	}
}

// thread_delow (prlHashReads.c:844-887)
template <int NW> __global__ __launch_bounds__(TPB) void k_delow(Table<NW> tbl, uint32_t d, Stats *stats)
{
	const uint64_t slots = tbl.slots();
	uint32_t removed = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY)
			continue;
		uint64_t v = tbl.ent[s].val;
		uint64_t nv = v;
#pragma unroll
		for (int f = 0; f < 8; f++) {
			const uint32_t c = (uint32_t)(v >> (6 * f)) & 63u;
			if (c > 0 && c <= d)
				nv &= ~(63ULL << (6 * f));
		}
		if (nv != v)
			tbl.ent[s].val = nv;
		if ((nv & 0xFFFFFFFFFFFFULL) == 0) {         // l_links == 0 && r_links == 0
			tbl.aux[s] |= AUX_DELETED;
			removed++;
		}
	}
#pragma unroll
	for (int dd = 32; dd > 0; dd >>= 1)
		removed += __shfl_down(removed, dd);
	if ((threadIdx.x & 63) == 0 && removed)
		atomicAdd(&stats->scratch, (unsigned long long)removed);
}

// thread_mark (prlHashReads.c:911-967): bins in LDS per workgroup, flushed once
template <int NW>
__global__ __launch_bounds__(TPB) void k_mark_hist(Table<NW> tbl, unsigned long long *__restrict__ hist, Stats *stats)
{
	__shared__ uint32_t s_hist[257];
	for (int i = threadIdx.x; i < 257; i += TPB)
		s_hist[i] = 0;
	__syncthreads();
	const uint64_t slots = tbl.slots();
	uint32_t linear = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY)
			continue;
		const uint64_t v = tbl.ent[s].val;
		const uint32_t aux = tbl.aux[s];
		uint32_t in_num = 0, out_num = 0, l_cvg = 0, r_cvg = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const uint32_t l = (uint32_t)(v >> (6 * b)) & 63u, r = (uint32_t)(v >> (24 + 6 * b)) & 63u;
			in_num += l > 0; l_cvg += l;
			out_num += r > 0; r_cvg += r;
		}
		const uint32_t count = ((aux & 0xFFFFu) << 16) | (uint32_t)(v >> 48);
		const uint32_t bin = count == 1 ? 1u : (l_cvg > r_cvg ? l_cvg : r_cvg);   // single <=> count == 1
		atomicAdd(&s_hist[bin], 1u);
		if (in_num == 1 && out_num == 1) {
			tbl.aux[s] = aux | AUX_LINEAR;
			linear++;
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < 257; i += TPB)
		if (s_hist[i])
			atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
#pragma unroll
	for (int dd = 32; dd > 0; dd >>= 1)
		linear += __shfl_down(linear, dd);
	if ((threadIdx.x & 63) == 0 && linear)
		atomicAdd(&stats->scratch, (unsigned long long)linear);
}

// compaction into kmer_t-shaped arrays (inc/newhash.h:65-77); order = arrival order of the cursor
template <int NW>
__global__ __launch_bounds__(TPB) void k_export(Table<NW> tbl, uint64_t *__restrict__ keys, uint32_t *__restrict__ l_links,
                                                uint32_t *__restrict__ r_flags, uint32_t *__restrict__ count,
                                                uint64_t *__restrict__ first, unsigned long long max_nodes, Stats *stats)
{
	const uint64_t slots = tbl.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY)
			continue;
		const unsigned long long pos = atomicAdd(&stats->scratch, 1ULL);   // hipcc aggregates this per wave
		if (pos >= max_nodes)
			continue;
		const uint32_t aux = tbl.aux[s];
		const uint32_t cnt = ((aux & 0xFFFFu) << 16) | (uint32_t)(e.val >> 48);
		if (keys) {
#pragma unroll
			for (int i = 0; i < NW; i++)
				keys[pos * NW + i] = e.key[i];
		}
		if (l_links) l_links[pos] = (uint32_t)(e.val & 0xFFFFFFu);
		if (r_flags)
			r_flags[pos] = (uint32_t)((e.val >> 24) & 0xFFFFFFu) | ((aux & AUX_LINEAR) ? 1u << 24 : 0u) |
			               ((aux & AUX_DELETED) ? 1u << 25 : 0u) | (cnt == 1 ? 1u << 27 : 0u);
		if (count) count[pos] = cnt;
		if (first) first[pos] = tbl.first ? tbl.first[s] : ORD_NONE;
	}
}

// growth: move every node of `src` into the (empty, larger) table `dst`; keys are unique so a claim is
// a plain CAS on the first word and the payload is copied, not re-counted
template <int NW> __global__ __launch_bounds__(TPB) void k_rehash(Table<NW> src, Table<NW> dst, Stats *stats)
{
	const uint64_t slots = src.slots();
	uint32_t failed = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = src.ent[s];
		if (e.key[0] == KEY_EMPTY)
			continue;
		Key<NW> key;
#pragma unroll
		for (int i = 0; i < NW; i++)
			key.w[i] = e.key[i];
		uint64_t slot = flat_home<NW>(dst, key);
		bool placed = false;
		for (uint64_t probe = 0; probe < dst.fslots; probe++) {
			const uint64_t old = atomicCAS((unsigned long long *)&dst.ent[slot].key[0], (unsigned long long)KEY_EMPTY,
			                               (unsigned long long)e.key[0]);
			if (old == KEY_EMPTY) {
#pragma unroll
				for (int i = 1; i < NW; i++)
					dst.ent[slot].key[i] = e.key[i];
				dst.ent[slot].val = e.val;
				dst.aux[slot] = src.aux[s];
				if (dst.first)
					dst.first[slot] = src.first[s];
				placed = true;
				break;
			}
			slot = flat_next(slot, dst.fslots);
		}
		if (!placed)
			failed++;
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

// nodes counted elsewhere (another rank's shard, sdt_gpu_export_nodes layout) become nodes of this table
template <int NW>
__global__ __launch_bounds__(TPB) void k_import(Table<NW> tbl, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ l_links,
                                                const uint32_t *__restrict__ r_flags, const uint32_t *__restrict__ count,
                                                const uint64_t *__restrict__ first, uint64_t n, Stats *stats)
{
	uint32_t claimed = 0, failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		Key<NW> key;
#pragma unroll
		for (int w = 0; w < NW; w++)
			key.w[w] = keys[i * NW + w];
		uint64_t slot, seen;
		const uint32_t before = claimed;
		if (!table_locate<NW>(tbl, key, claimed, slot, seen) || claimed == before) {
			failed++;                                // no room, or the key is already there: shards are disjoint
			continue;
		}
		const uint32_t rf = r_flags[i], cnt = count[i];
		tbl.ent[slot].val = ((uint64_t)(cnt & 0xFFFFu) << 48) | ((uint64_t)(rf & 0xFFFFFFu) << 24) | (uint64_t)(l_links[i] & 0xFFFFFFu);
		tbl.aux[slot] = (cnt >> 16) | ((rf >> 24) & 1u ? AUX_LINEAR : 0u) | ((rf >> 25) & 1u ? AUX_DELETED : 0u);
		if (tbl.first)
			tbl.first[slot] = first ? first[i] : ORD_NONE;
	}
	if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

// the final graph as the second read pass needs it -- key -> path word -- out of one rank's table and into another's (--gpus N: every rank
// maps its own reads, prlRead2path.c:817-1335 on every rank's share of the input)
template <int NW>
__global__ __launch_bounds__(TPB) void k_export_paths(Table<NW> tbl, uint64_t *__restrict__ keys, uint64_t *__restrict__ paths, unsigned long long max_nodes, Stats *stats)
{
	const uint64_t slots = tbl.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY)
			continue;
		const unsigned long long pos = atomicAdd(&stats->scratch, 1ULL);
		if (pos >= max_nodes)
			continue;
#pragma unroll
		for (int i = 0; i < NW; i++)
			keys[pos * NW + i] = e.key[i];
		paths[pos] = e.val;
	}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_import_paths(Table<NW> tbl, const uint64_t *__restrict__ keys, const uint64_t *__restrict__ paths, uint64_t n, Stats *stats)
{
	uint32_t claimed = 0, failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		Key<NW> key;
#pragma unroll
		for (int w = 0; w < NW; w++)
			key.w[w] = keys[i * NW + w];
		uint64_t slot, seen;
		const uint32_t before = claimed;
		if (!table_locate<NW>(tbl, key, claimed, slot, seen) || claimed == before) {
			failed++;                                // no room, or the key twice
			continue;
		}
		tbl.ent[slot].val = paths[i];
	}
	if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

#include "sdt_superkmer_kernels.cuh"
#include "sdt_comm.cuh"
#include "sdt_shard_plan.h"
#include "sdt_count_plan.h"
static_assert(SHARD_NB1 == SK_NB1, "the exchange plan and the pipeline agree about the level-1 buckets");
#include "sdt_append.cuh"
#include "sdt_map_kernels.cuh"
#include "sdt_ctg_kernels.cuh"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct EventPair {
	hipEvent_t a, b;
	uint64_t kmers;
	int stage;               // SDT_STAGE_*
};

struct sdt_ctx {
	int device = 0;
	int K = 0;
	int nw = 1;
	uint64_t slots = 0;
	void *d_ent = nullptr;
	uint32_t *d_aux = nullptr;
	uint64_t *d_first = nullptr;       // SDT_FLAG_TRACK_FIRST
	uint64_t ord_base = 0, ord_stride = 1;
	Stats *d_stats = nullptr;
	Stats *h_stats = nullptr;          // pinned
	unsigned long long *d_hist = nullptr;
	hipStream_t stream = nullptr, copy_stream = nullptr;
	bool own_stream = true;
	// host-batch staging (double buffered)
	// A ring of NSTAGE device buffers.  A pushed batch is COPIED at once (copy stream) and queued; its kernels are launched
	// from the queue.  The one thing that blocks the host for long is a flush of the locality pipeline (two host syncs for
	// the chunk lists: ~30 ms per 2^31 k-mers), and while the host is blocked nobody feeds the copy engine -- so a launch
	// that needs a flush is put off until STAGE_AHEAD copies are queued behind it: the copies then run while the host waits
	// (without this the PCIe-inclusive rate was compute + copy, not max(compute, copy): 50 vs 76 G k-mers/s resident).
	static constexpr int NSTAGE = 48, STAGE_AHEAD = 32;
	uint32_t *d_words[NSTAGE] = {};
	uint64_t *d_offs[NSTAGE] = {};
	uint64_t cap_words[NSTAGE] = {}, cap_offs[NSTAGE] = {};
	hipEvent_t buf_free[NSTAGE] = {}, copied[NSTAGE] = {};
	int next_buf = 0;
	struct Staged { const uint32_t *dw; uint64_t *dof; uint64_t nreads, maxlen, ord_base, ord_stride; int slot; uint64_t fixed_len; };
	std::vector<Staged> staged;        // copied (or being copied), not yet launched: [staged_head, size)
	size_t staged_head = 0;
	bool draining = false;
	uint64_t push_ord_base = 0, push_ord_stride = 1;      // ordinals of the next PUSHED batch (ord_base / ord_stride: of the next LAUNCHED one)
	uint64_t push_ticket = 0;          // pushes issued so far: ticket t's host buffers are free once copied[(t - 1) % NSTAGE] has passed
	uint64_t expect_kmers = 0;         // sdt_gpu_hint_total_kmers
	// bookkeeping for growth: upper bound of distinct nodes without syncing
	uint64_t distinct_known = 0;       // as of the last sync
	uint64_t kmers_known = 0;          // occurrences counted as of the last sync (new nodes per occurrence: bound of the next launch)
	uint64_t kmers_since_sync = 0;     // launched since then (an upper bound of the new nodes they may bring)
	uint64_t hard_since_sync = 0;      // k-mers launched since then by the locality pipeline, whatever its own bound said
	uint64_t kmers_total_host = 0;
	uint64_t kmers_offered = 0;        // upper bound of the k-mers handed to pass 1 since the last reset (picks the kernel family)
	uint32_t flags = 0;
	// locality pipeline (sdt_superkmer.cuh): chunk pools of the two scatter levels, chunk lists, pending work
	struct SkState {
		bool ready = false;
		uint64_t cap_kmers = 0;            // k-mers the pools are sized for (one batch)
		bool cap_is_max = false;           // the device has no room for larger pools: do not try again
		uint64_t pending_kmers = 0;        // scattered into pool 1, not yet counted
		SkPool p1 = {nullptr, nullptr, nullptr, 0}, p2 = {nullptr, nullptr, nullptr, 0};
		unsigned long long *cursors = nullptr;   // [wgs][SK_NB1] open chunks of the level-1 scatter
		unsigned long long *blk = nullptr;       // [wgs] block of chunk ids each workgroup is handing out
		uint32_t wgs = 0;
		uint32_t *cnt1 = nullptr, *off1 = nullptr, *fill1 = nullptr, *list1 = nullptr;
		uint32_t *cnt2 = nullptr, *off2 = nullptr, *fill2 = nullptr, *list2 = nullptr;
		unsigned long long *kmers2 = nullptr, *kpre2 = nullptr;
		SkItem *items = nullptr;
		uint32_t items_cap = 0;
		uint4 *citems = nullptr, *h_citems = nullptr;       // work items of k_sk_count: [c0, c1) in list2 + their final buckets (device / pinned; sdt_count_plan.h)
		uint32_t citems_cap = 0;
		uint32_t *next_item = nullptr;                      // one counter per k_sk_count launch
		uint32_t *h_off1 = nullptr, *h_off2 = nullptr;      // pinned
		unsigned long long *h_kpre2 = nullptr;              // pinned
		SkItem *h_items = nullptr;                          // pinned
		bool flushing = false;
		// statistics of the last flush (sdt_gpu_pipeline_stats)
		uint64_t st_records = 0, st_chunks1 = 0, st_chunks2 = 0, st_flushes = 0;
		uint32_t stream_flushes = 0;   // flushes since the last reset (sk_batch_limit)
		uint64_t l2_in_total = 0;      // k-mers that entered the count stage (sum of the level-2 bucket sizes): Stats.sk_counted must match
		bool exchanged = false;        // records left for / came from other ranks: Stats.sk_emitted is not this rank's input
	} sk;
	// multi-GPU (sdt_comm.cuh): communicator + double-buffered send / receive chunk buffers of the exchange
	Comm comm;
	struct Shard {
		uint64_t *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};       // chunk payloads
		uint32_t *send_meta[2] = {nullptr, nullptr}, *recv_meta[2] = {nullptr, nullptr};
		uint32_t *iota = nullptr;                                                    // identity chunk list of a receive buffer
		uint32_t send_chunks = 0, recv_chunks = 0;
		hipEvent_t ev_gather[2] = {nullptr, nullptr}, ev_xdone[2] = {nullptr, nullptr}, ev_l2[2] = {nullptr, nullptr};
		bool x_recorded[2] = {false, false}, l2_recorded[2] = {false, false};
		uint64_t round = 0;
		// what the last exchange delivered and sk_split has not consumed yet
		bool pending = false;
		int pending_slot = 0;
		uint32_t pending_items = 0;
		std::vector<SkItem> items;
		uint64_t kmers_scattered = 0;
		// which rank owns which level-1 buckets: contiguous ranges [ranges[r], ranges[r + 1]), balanced by the bucket
		// weights of a sample of the first call's reads (the same on every rank: the weights are all-gathered)
		bool have_ranges = false;
		uint32_t ranges[65] = {0};
	} sh;
	// second pass (prlRead2edge): reads kept from pass 1, path words, patch table, arcs
	struct KeptBatch { uint32_t *d_words; uint64_t *d_offs; uint64_t nwords, nreads, ord_base, ord_stride, maxlen; };
	std::vector<KeptBatch> kept;
	// kept batches live in a few large slabs (two hipMallocs per 32 MiB batch were thousands of synchronous calls on
	// the ingest path): bump allocation, everything is released together
	struct KeepSlab { uint8_t *p; size_t size, used; };
	std::vector<KeepSlab> keep_slabs;
	void *d_patch = nullptr;
	uint64_t patch_slots = 0;
	ArcEnt *d_arcs = nullptr;
	uint64_t arc_slots = 0;
	bool paths_loaded = false;
	uint64_t *d_idx = nullptr;         // slot -> index of the node in the host's visiting order (sdt_gpu_layout_apply / sdt_gpu_set_node_index)
	sdti::GraphExt *gx = nullptr;      // graph phases (sdt_gpu_graph.hip)
	uint64_t idx_slots = 0, idx_n = 0;
	// map stage (SDT_FLAG_CONTIG_INDEX): contig ordinal -> id, contig_array, staging for sdt_gpu_align_reads
	uint32_t *d_ctg_ids = nullptr;
	uint64_t ctg_ord = 0, ctg_ids_cap = 0;
	uint32_t *d_ctg_len = nullptr, *d_ctg_twin = nullptr;
	uint64_t num_ctg = 0;
	void *ab[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // words, offsets, align_len, read_info, hits
	size_t ab_cap[5] = {0, 0, 0, 0, 0};
	unsigned long long *d_hit_cursor = nullptr;
	bool index_final = false;            // k_finalize_contig_index has run: look-ups only from here on
	// timing
	std::vector<EventPair> ev;
	size_t ev_used = 0;
	int cu_count = 256;
};

static const double MAX_LOAD = 0.70;
// slots of a flat table for `nodes` nodes: a load of SDT_TABLE_LOAD percent (default 45: measured best on the headline workload, between what the probes of the merges
// like and what the scans of the table cost), a multiple of 4096, 2^16 at least
static uint64_t flat_slots_for(uint64_t nodes)
{
	static const int pct = getenv("SDT_TABLE_LOAD") && atoi(getenv("SDT_TABLE_LOAD")) >= 10 && atoi(getenv("SDT_TABLE_LOAD")) <= 69 ? atoi(getenv("SDT_TABLE_LOAD")) : 45;
	uint64_t slots = (uint64_t)((double)nodes * 100.0 / pct) + 4095;
	slots &= ~4095ULL;
	return slots < (1ULL << 16) ? (1ULL << 16) : slots;
}

static void *keep_alloc(sdt_ctx *c, size_t bytes)
{
	bytes = (bytes + 255) & ~(size_t)255;
	if (c->keep_slabs.empty() || c->keep_slabs.back().size - c->keep_slabs.back().used < bytes) {
		sdt_ctx::KeepSlab sl;
		sl.size = bytes > ((size_t)1 << 30) ? bytes : ((size_t)1 << 30);
		sl.used = 0;
		sl.p = nullptr;
		if (hipMalloc((void **)&sl.p, sl.size) != hipSuccess) {
			sl.size = bytes;                             // a full slab does not fit any more: exactly what is needed
			if (hipMalloc((void **)&sl.p, sl.size) != hipSuccess)
				return nullptr;
		}
		c->keep_slabs.push_back(sl);
	}
	sdt_ctx::KeepSlab &b = c->keep_slabs.back();
	void *r = b.p + b.used;
	b.used += bytes;
	return r;
}

static void keep_release(sdt_ctx *c)
{
	for (auto &sl : c->keep_slabs) (void)hipFree(sl.p);
	c->keep_slabs.clear();
	c->kept.clear();
}

// the flat table: what the direct kernel family counts into (and grows)
template <int NW> static Table<NW> flat_of(const sdt_ctx *c)
{
	Table<NW> t;
	t.ent = (Entry<NW> *)c->d_ent;
	t.aux = c->d_aux;
	t.fslots = c->slots;
	t.first = c->d_first;
	return t;
}

// the node table as every stage after pass 1 sees it
template <int NW> static Table<NW> table_of(const sdt_ctx *c) { return flat_of<NW>(c); }
static uint64_t view_slots(const sdt_ctx *c) { return c->slots; }

static size_t entry_bytes(int nw) { return nw == 1 ? sizeof(Entry<1>) : nw == 2 ? sizeof(Entry<2>) : sizeof(Entry<4>); }

static int scan_grid(const sdt_ctx *c, uint64_t items) { return sdti::scan_grid(c->cu_count, items); }

static int launch_clear(sdt_ctx *c, void *ent, uint32_t *aux, uint64_t *first, uint64_t slots)
{
	const int g = scan_grid(c, slots);
	if (c->nw == 1) { Table<1> t{(Entry<1> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<1>, dim3(g), dim3(TPB), 0, c->stream, t); }
	else if (c->nw == 2) { Table<2> t{(Entry<2> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<2>, dim3(g), dim3(TPB), 0, c->stream, t); }
	else { Table<4> t{(Entry<4> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<4>, dim3(g), dim3(TPB), 0, c->stream, t); }
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

static int alloc_table(sdt_ctx *c, uint64_t slots, void **ent, uint32_t **aux, uint64_t **first)
{
	*ent = nullptr;
	*aux = nullptr;
	*first = nullptr;
	hipError_t e = hipMalloc(ent, slots * entry_bytes(c->nw));
	if (e == hipSuccess)
		e = hipMalloc((void **)aux, slots * sizeof(uint32_t));
	if (e == hipSuccess && (c->flags & SDT_FLAG_TRACK_FIRST))
		e = hipMalloc((void **)first, slots * sizeof(uint64_t));
	if (e != hipSuccess) {
		if (*ent) (void)hipFree(*ent);
		if (*aux) (void)hipFree(*aux);
		*ent = nullptr;
		*aux = nullptr;
		return fail(SDT_ENOMEM, "node table: hipMalloc(%llu slots x %zu B) failed: %s", (unsigned long long)slots,
		            entry_bytes(c->nw) + 4, hipGetErrorString(e));
	}
	return SDT_OK;
}

static int sk_flush(sdt_ctx *c);
static void sk_free(sdt_ctx *c);

static int env_int(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }
static int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static EventPair *next_event(sdt_ctx *c);
static int drain_staged(sdt_ctx *c, bool force);

static int sync_stats(sdt_ctx *c)
{
	// batches that were pushed but not launched yet, and work parked in the locality pipeline, belong to the table before
	// anybody looks at it
	if (!c->draining) {
		const int rcd = drain_staged(c, true);
		if (rcd != SDT_OK)
			return rcd;
	}
	if (c->sk.pending_kmers && !c->sk.flushing) {
		const int rcf = sk_flush(c);
		if (rcf != SDT_OK)
			return rcf;
	}
	HIPCHK(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
	{ const int rcw = c->comm.sync_watched(c->stream, "the drain of pass 1"); if (rcw != SDT_OK) return rcw; }
	if (c->h_stats->probe_fail)
		return fail(SDT_EFULL, "%llu inserts found no slot (table over-full or route bucket overflow)",
		            (unsigned long long)c->h_stats->probe_fail);
	// conservation in the locality pipeline: every k-mer cut into a record must come out of the count stage.  (A
	// workgroup geometry that lost a chunk now and then was found with exactly this comparison; it costs two counters.)
	if (!c->sk.flushing) {
		if (c->h_stats->sk_counted != c->sk.l2_in_total)
			return fail(SDT_ESTATE, "locality pipeline lost k-mers: %llu entered the count stage, %llu were counted",
			            (unsigned long long)c->sk.l2_in_total, (unsigned long long)c->h_stats->sk_counted);
		if (!c->sk.exchanged && c->h_stats->sk_emitted != c->sk.l2_in_total)
			return fail(SDT_ESTATE, "locality pipeline lost k-mers: %llu were cut into records, %llu reached the count stage",
			            (unsigned long long)c->h_stats->sk_emitted, (unsigned long long)c->sk.l2_in_total);
	}
	c->distinct_known = c->h_stats->distinct;
	c->kmers_known = c->h_stats->kmers;
	c->kmers_since_sync = 0;
	c->hard_since_sync = 0;
	return SDT_OK;
}

static int grow_table(sdt_ctx *c, uint64_t need_nodes)
{
	// (any number of slots: what the nodes need at the load a fresh table is sized for, at least half as many again as before)
	uint64_t slots = flat_slots_for(need_nodes);
	if (slots < c->slots + c->slots / 2) slots = c->slots + c->slots / 2;
	void *ent = nullptr;
	uint32_t *aux = nullptr;
	uint64_t *first = nullptr;
	int rc = alloc_table(c, slots, &ent, &aux, &first);
	if (rc != SDT_OK && c->sk.ready && !c->sk.flushing) {
		// the locality pipeline's pools are only a cache of work: count what they hold, give the memory back, try again
		rc = sk_flush(c);
		if (rc == SDT_OK) {
			HIPCHK(hipStreamSynchronize(c->stream));
			sk_free(c);
			rc = alloc_table(c, slots, &ent, &aux, &first);
		}
	}
	if (rc != SDT_OK)
		return fail(SDT_EFULL, "cannot grow node table to %llu slots: %s", (unsigned long long)slots, g_err);
	rc = launch_clear(c, ent, aux, first, slots);
	if (rc != SDT_OK)
		return rc;
	const int g = scan_grid(c, c->slots);
	if (c->nw == 1) { Table<1> d{(Entry<1> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d, c->d_stats); }
	else if (c->nw == 2) { Table<2> d{(Entry<2> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d, c->d_stats); }
	else { Table<4> d{(Entry<4> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d, c->d_stats); }
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipFree(c->d_ent));
	HIPCHK(hipFree(c->d_aux));
	if (c->d_first) HIPCHK(hipFree(c->d_first));
	c->d_ent = ent;
	c->d_aux = aux;
	c->d_first = first;
	c->slots = slots;
	return SDT_OK;
}

// make sure `incoming` more occurrences cannot push the table past MAX_LOAD (every occurrence might
// be a new node); syncs only when the cheap upper bound says it could
static int ensure_room(sdt_ctx *c, uint64_t incoming)
{
	const double room = (double)c->slots * MAX_LOAD;
	if ((double)(c->distinct_known + c->kmers_since_sync + incoming) <= room)
		return SDT_OK;
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if ((double)(c->distinct_known + incoming) <= room)
		return SDT_OK;
	return grow_table(c, c->distinct_known + incoming);
}

static EventPair *next_event(sdt_ctx *c)
{
	if (c->ev_used == c->ev.size()) {
		EventPair p;
		if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)
			return nullptr;
		p.kmers = 0;
		p.stage = 0;
		c->ev.push_back(p);
	}
	c->ev[c->ev_used].stage = 0;
	return &c->ev[c->ev_used++];
}

static size_t tile_smem_bytes(int max_tile_words)
{
	return (size_t)(2 * (TILE_READS + 1) + 2 + LDS_LEAD + max_tile_words) * sizeof(uint32_t);
}

// the largest tile a batch can produce: TILE_READS consecutive reads; computed on the host side from the
// maximum read length the caller promised (offsets are device resident for the device entry point)
static int tile_words_for(uint64_t max_read_len)
{
	const uint64_t bases = (uint64_t)TILE_READS * max_read_len + 16;
	return (int)((bases + 15) / 16) + TAIL_PAD + 1;
}

// ------------------------------------------------------------------------------------------------
// locality pipeline (sdt_superkmer.cuh): scatter super-k-mers -> split -> count in LDS -> merge
// ------------------------------------------------------------------------------------------------
static const uint64_t SK_BATCH_MAX_KMERS = 1ULL << clamp_int(env_int("SDT_SK_BATCH_LOG2", 34), 24, 36);      // k-mers per batch at most (pools: ~6 B per k-mer at K = 31)
static const uint32_t SK_ITEM_CHUNKS = 4096;                // level-1 chunks per level-2 work item (4 MiB of records)
static const uint64_t SK_COUNT_KMERS = 1ULL << clamp_int(env_int("SDT_SK_COUNT_KMERS_LOG2", 29), 20, 36);
static const uint32_t SK_COUNT_PACK_CHUNKS = 64;            // level-2 chunks up to which neighbouring small buckets share a work item (1 K records = two tiles)
static const uint32_t SK_COUNT_ITEM_CHUNKS = (uint32_t)clamp_int(env_int("SDT_SK_COUNT_ITEM_CHUNKS", 1024), 64, 1 << 24);          // level-2 chunks per k_sk_count work item (16 K records); a bucket within it is counted by ONE workgroup (owned merges)
static const uint32_t SK_MAX_COUNT_LAUNCHES = 4096;          // k-mers per k_sk_count launch (growth bound, see ensure_room)

static void sk_free(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	void *dev[] = {k.p1.recs, k.p1.meta, k.p1.next, k.p2.recs, k.p2.meta, k.p2.next, k.cursors, k.blk, k.cnt1, k.off1, k.fill1, k.list1,
	               k.cnt2, k.off2, k.fill2, k.list2, k.kmers2, k.kpre2, k.items, k.citems, k.next_item};
	for (void *p : dev)
		if (p) (void)hipFree(p);
	void *host[] = {k.h_off1, k.h_off2, k.h_kpre2, k.h_items, k.h_citems};
	for (void *p : host)
		if (p) (void)hipHostFree(p);
	const uint64_t in_total = k.l2_in_total;         // (the conservation totals belong to the run, not to the pools)
	const bool exchanged = k.exchanged;
	const uint32_t stream_flushes = k.stream_flushes;
	k = sdt_ctx::SkState();
	k.l2_in_total = in_total;
	k.exchanged = exchanged;
	k.stream_flushes = stream_flushes;
}

// LDS bytes of the level-1 scatter for a maximum read length
struct SkGeo { int mtw, tile_words, hv_words, hv2_words, bits_words; size_t smem; };
static SkGeo sk_geo(int K, uint64_t max_read_len)
{
	SkGeo g;
	g.mtw = (int)(((uint64_t)SK_TILE_READS * max_read_len + 16 + 15) / 16) + TAIL_PAD + 1;
	g.tile_words = (int)((tile_smem_bytes(g.mtw) / sizeof(uint32_t) + 1) & ~(size_t)1);
	g.hv_words = (int)((SK_TILE_READS * max_read_len + 16 + 1) & ~(uint64_t)1);
	const uint64_t nk_max = (uint64_t)SK_TILE_READS * (max_read_len - K + 1);
	g.bits_words = (int)(nk_max / 64 + 2);
	// long windows (K - m + 1 > 49: the strip kernel's sparse table of window minima) ping-pong between two hash arrays
	g.hv2_words = K - sk_minimizer_len(K) + 1 > 49 ? g.hv_words : 0;
	g.smem = (size_t)g.tile_words * 4 + (size_t)SK_NB1 * 8 + (size_t)(g.hv_words + g.hv2_words) * 4 + (size_t)g.bits_words * 8 + (size_t)(g.bits_words + 2) * 4;
	return g;
}

template <int NW, bool TRACK> static size_t sk_count_smem()
{
	using G = SkCntGeo<NW, TRACK>;
	constexpr int SLOTS = G::SLOTS, BW = SkFmt<NW>::BW, TR = G::TILE;
	// keys (+ ordinals), headers, 5 field words per slot, weights, the prefix / map / index region (= dedupe table), the tile's bases
	return (size_t)(NW + (TRACK ? 1 : 0)) * SLOTS * 8 + (size_t)TR * 8 + (size_t)SLOTS * 20 + (size_t)TR * 4 + G::REGION +
	       (size_t)(LDS_LEAD + TR * BW * 2 + TAIL_PAD) * 4;
}

static bool sk_applicable(const sdt_ctx *c, uint64_t max_read_len)
{
	if (c->flags & SDT_FLAG_CONTIG_INDEX)
		return false;
	if (max_read_len < (uint64_t)c->K + 1 || max_read_len > (uint64_t)SK_MAX_READ_LEN)
		return false;
	return sk_geo(c->K, max_read_len).smem <= 160 * 1024;
}

// pool 1 empty, every workgroup without an open chunk
static int sk_reset_pool1(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	HIPCHK(hipMemsetAsync(k.p1.next, 0, 4, c->stream));
	HIPCHK(hipMemsetAsync(k.cnt1, 0, SK_NB1 * 4, c->stream));
	hipLaunchKernelGGL(k_sk_init_cursors, dim3(256), dim3(256), 0, c->stream, k.cursors, k.wgs * (uint32_t)SK_NB1, (uint32_t)SK_CAP1, k.blk, k.wgs);
	HIPCHK(hipGetLastError());
	k.pending_kmers = 0;
	return SDT_OK;
}

static int sk_alloc(sdt_ctx *c, uint64_t want_kmers, uint64_t per_read)
{
	sdt_ctx::SkState &k = c->sk;
	if (want_kmers > SK_BATCH_MAX_KMERS)
		want_kmers = SK_BATCH_MAX_KMERS;
	if (k.ready && (k.cap_kmers >= want_kmers || k.cap_is_max || k.pending_kmers))
		return SDT_OK;                               // (pools that hold records are never replaced: they are flushed first)
	if (k.ready) {
		HIPCHK(hipStreamSynchronize(c->stream));
		sk_free(c);
	}
	const int rw = sk_rec_words(c->nw), rw2 = sk_rec2_stride(c->nw);      // words per record; per slot of a level-2 chunk
	const int w = c->K - sk_minimizer_len(c->K) + 1;
	// records: a run ends where the minimizer's bucket changes (every (w + 1) / 2 k-mers for a random order of the m-mers) or
	// where the record is full (every `max run` k-mers at the latest): 1 / (2 / (w + 1) + 1 / max run) k-mers per record is what
	// the pools are sized for.  Measured: 10.2 k-mers per record against 8.2 from this formula at K = 31, 23.9 against 19 at
	// K = 63, 6.5 against 5.7 at K = 23 -- the margin IS the head room (a record that finds no chunk takes the direct path,
	// `pool_direct` in the pipeline statistics: nothing is lost, but a fifth of the k-mers of a K = 95 job went that way and
	// tripled its scatter time when the pools were sized at (w + 1) / 2 * 3 / 4).  4-word keys and reads of more than 256 k-mers go
	// through the strip kernel, which also cuts at multiples of the record capacity: a fifth more room.  A read is at least one record.
	const double rate = 2.0 / (double)(w + 1) + 1.0 / (double)sk_max_run(c->K, c->nw);
	double run = 1.0 / rate * ((c->nw == 4 || per_read > (uint64_t)SK_SEQ_MAX_KMERS) ? 0.8 : 1.0);
	if (run > (double)per_read) run = (double)per_read;
	uint64_t div = (uint64_t)run;
	if (div < 2) div = 2;
	div = (uint64_t)clamp_int(env_int("SDT_SK_POOL_DIV", (int)div), 1, 1 << 20);
	const int mem_pct = env_int("SDT_SK_POOL_MEM_PCT", 60);
	size_t free_b = 0, total_b = 0;
	HIPCHK(sdti::mem_info(&free_b, &total_b));
	uint64_t cap = want_kmers < (1ULL << 24) ? (1ULL << 24) : want_kmers;
	k.cap_is_max = cap >= SK_BATCH_MAX_KMERS;
	const uint32_t wgs = (uint32_t)c->cu_count * 6;
	for (;; cap /= 2) {
		const uint64_t recs = cap / div;
		// (SDT_SK_POOL_CHUNKS1: test hook -- a level-1 pool of that many chunks, so that small inputs overflow it: tests/test_gpu_parity.py,
		// tests/test_sharded.py)
		const uint64_t chunks1 = env_int("SDT_SK_POOL_CHUNKS1", 0) > 0 ? (uint64_t)env_int("SDT_SK_POOL_CHUNKS1", 0) : recs / SK_CAP1 + (uint64_t)wgs * SK_NB1 + 1024;
		const uint64_t items = chunks1 / SK_ITEM_CHUNKS + SK_NB1 + 1;
		const uint64_t chunks2 = chunks1 * (SK_CAP1 / SK_CAP2) + items * SK_NB2 + 1024;
		const uint64_t bytes = chunks1 * SK_CAP1 * rw * 8 + chunks2 * SK_CAP2 * rw2 * 8 + (chunks1 + chunks2) * 8;
		// (a sharded context adds two send and two receive buffers of pool-1 size: shard_alloc)
		const uint64_t all = c->comm.nranks > 1 ? bytes + chunks1 * SK_CAP1 * rw * 8 * 9 / 2 : bytes;
		if (chunks2 >= (1ULL << SK_LIST2_FILL_SHIFT) - 1 || all > free_b / 100 * (uint64_t)mem_pct) {      // (a list2 entry has 28 bits for the chunk id; all ones = no chunk)
			k.cap_is_max = true;
			if (cap <= (1ULL << 24))
				return fail(SDT_ENOMEM, "super-k-mer pools: %llu MiB needed for the smallest batch, %zu MiB free",
				            (unsigned long long)(bytes >> 20), free_b >> 20);
			continue;
		}
		k.p1.chunks = (uint32_t)chunks1;
		k.p2.chunks = (uint32_t)chunks2;
		k.items_cap = (uint32_t)items;
		break;
	}
	k.wgs = wgs;
	const double t_alloc0 = comm_now();
	HIPCHK(hipMalloc((void **)&k.p1.recs, (size_t)k.p1.chunks * SK_CAP1 * rw * 8));
	HIPCHK(hipMalloc((void **)&k.p1.meta, (size_t)k.p1.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.p1.next, 64));
	HIPCHK(hipMalloc((void **)&k.p2.recs, (size_t)k.p2.chunks * SK_CAP2 * rw2 * 8));
	HIPCHK(hipMalloc((void **)&k.p2.meta, (size_t)k.p2.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.p2.next, 64));
	HIPCHK(hipMalloc((void **)&k.cursors, (size_t)wgs * SK_NB1 * 8));
	HIPCHK(hipMalloc((void **)&k.blk, (size_t)wgs * 8));
	HIPCHK(hipMalloc((void **)&k.cnt1, SK_NB1 * 4));
	HIPCHK(hipMalloc((void **)&k.off1, (SK_NB1 + 1) * 4));
	HIPCHK(hipMalloc((void **)&k.fill1, SK_NB1 * 4));
	HIPCHK(hipMalloc((void **)&k.list1, (size_t)k.p1.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.cnt2, SK_NBF * 4));
	HIPCHK(hipMalloc((void **)&k.off2, (SK_NBF + 1) * 4));
	HIPCHK(hipMalloc((void **)&k.fill2, SK_NBF * 4));
	HIPCHK(hipMalloc((void **)&k.list2, (size_t)k.p2.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.kmers2, SK_NBF * 8));
	HIPCHK(hipMalloc((void **)&k.kpre2, (SK_NBF + 1) * 8));
	HIPCHK(hipMalloc((void **)&k.items, (size_t)k.items_cap * sizeof(SkItem)));
	HIPCHK(hipHostMalloc((void **)&k.h_off1, (SK_NB1 + 1) * 4, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_off2, (SK_NBF + 1) * 4, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_kpre2, (SK_NBF + 1) * 8, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_items, (size_t)k.items_cap * sizeof(SkItem), hipHostMallocDefault));
	k.citems_cap = (uint32_t)SK_NBF + k.p2.chunks / SK_COUNT_ITEM_CHUNKS + 1;
	HIPCHK(hipMalloc((void **)&k.citems, (size_t)k.citems_cap * sizeof(uint4)));
	HIPCHK(hipHostMalloc((void **)&k.h_citems, (size_t)k.citems_cap * sizeof(uint4), hipHostMallocDefault));
	HIPCHK(hipMalloc((void **)&k.next_item, SK_MAX_COUNT_LAUNCHES * sizeof(uint32_t)));
	k.cap_kmers = cap;
	k.ready = true;
	if (getenv("SDT_TIMING"))
		fprintf(stderr, "[libsdt_gpu] super-k-mer pools for %llu k-mers per batch: %.1f GiB in %.0f ms\n", (unsigned long long)cap,
		        ((double)k.p1.chunks * SK_CAP1 * rw + (double)k.p2.chunks * SK_CAP2 * rw2) * 8 / (1 << 30), (comm_now() - t_alloc0) * 1e3);
	return sk_reset_pool1(c);
}

template <int NW, bool TRACK> static int sk_launch_count_t(sdt_ctx *c, uint32_t i0, uint32_t i1, uint32_t launch)
{
	sdt_ctx::SkState &k = c->sk;
	const size_t smem = sk_count_smem<NW, TRACK>();
	// persistent workgroups: as many as the LDS tables let the chip hold; they take work items first come first served
	const unsigned per_cu = (unsigned)((160 * 1024) / (smem + 256));
	unsigned grid = (unsigned)c->cu_count * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
	if (grid > i1 - i0) grid = i1 - i0;
	HIPCHK(hipFuncSetAttribute((const void *)k_sk_count<NW, TRACK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
	hipLaunchKernelGGL((k_sk_count<NW, TRACK>), dim3(grid), dim3(SkCntGeo<NW, TRACK>::TPB), smem, c->stream, k.p2, k.list2, (const uint4 *)k.citems, i0, i1,
	                   k.next_item + launch, c->K, flat_of<NW>(c), c->d_stats);
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

template <int NW> static int sk_launch_count(sdt_ctx *c, uint32_t i0, uint32_t i1, uint32_t launch)
{
	return c->d_first ? sk_launch_count_t<NW, true>(c, i0, i1, launch) : sk_launch_count_t<NW, false>(c, i0, i1, launch);
}

#define SK_CHK(expr)                                                                                   \
	do {                                                                                               \
		hipError_t e4_ = (expr);                                                                       \
		if (e4_ != hipSuccess)                                                                         \
			return fail(e4_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e4_), __FILE__, __LINE__); \
	} while (0)

// level 1 done: close the open chunks, list the chunk ids bucket by bucket; h_off1 is valid on return (host sync)
static int sk_list1(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	const int g = c->cu_count * 8;
	hipLaunchKernelGGL(k_sk_seal, dim3(256), dim3(256), 0, c->stream, k.cursors, k.wgs * (uint32_t)SK_NB1, k.blk, k.wgs, k.p1, (uint32_t)SK_CAP1);
	hipLaunchKernelGGL(k_sk_scan, dim3(1), dim3(1024), 0, c->stream, k.cnt1, k.off1, k.fill1, (int)SK_NB1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
	hipLaunchKernelGGL(k_sk_chunk_place_few, dim3(g), dim3(256), 0, c->stream, k.p1, k.off1, k.fill1, k.list1, (int)SK_NB1);
	SK_CHK(hipGetLastError());
	SK_CHK(hipMemcpyAsync(k.h_off1, k.off1, (SK_NB1 + 1) * 4, hipMemcpyDeviceToHost, c->stream));
	{ const int rcw = c->comm.sync_watched(c->stream, "the chunk lists of a round"); if (rcw != SDT_OK) return rcw; }
	k.st_chunks1 = k.h_off1[SK_NB1];
	return SDT_OK;
}

// level 2: the nitems work items in k.h_items (runs of chunks of `src` named by `list`) are split into pool 2, whose chunks
// are then listed per final bucket; asynchronous (the lists are read back by sk_count_all).  `after_l2`: recorded once
// the records have left `src`.
static int sk_split(sdt_ctx *c, const SkPool &src, const uint32_t *list, uint32_t nitems, hipEvent_t after_l2)
{
	sdt_ctx::SkState &k = c->sk;
	const int g = c->cu_count * 8;
	SK_CHK(hipMemsetAsync(k.p2.next, 0, 4, c->stream));
	SK_CHK(hipMemsetAsync(k.kmers2, 0, SK_NBF * 8, c->stream));
	SK_CHK(hipMemsetAsync(k.cnt2, 0, SK_NBF * 4, c->stream));
	if (nitems) {
		SK_CHK(hipMemcpyAsync(k.items, k.h_items, (size_t)nitems * sizeof(SkItem), hipMemcpyHostToDevice, c->stream));
		// ONE workgroup (512 lanes) per CU -- the unused dynamic LDS is there to keep a second one away.  Every workgroup has
		// 1024 chunks open and a record is 24..56 bytes of a 128-byte line: with 8 x 256 lanes per CU the lines being
		// filled (2 M of them, 270 MB) did not live in L2 until they were full and reached HBM as partial writes
		// (22.3 ms per 6 G k-mers); 256 K open lines do (17.8 ms).  SDT_SK_L2_PAD_KB: A/B switch.  (1024 lanes per
		// workgroup were 1 ms faster still and lost a chunk of records in half of the runs of the hot-bucket test --
		// 512 and 256 never did in the same stress; sync_stats' conservation check is the net under this.)
		static const bool l2_old = getenv("SDT_SK_L2_OLD") != NULL;      // A/B switch: the round-2 kernel (a store per record into one of 1024 open chunks)
		if (!l2_old) {
			// staged (round 5): records wait in LDS for a group of 4 (2), groups are stored whole; one workgroup per CU by its LDS alone
			const size_t sm = c->nw == 1 ? SkL2Stage<1>::SMEM : (c->nw == 2 ? SkL2Stage<2>::SMEM : SkL2Stage<4>::SMEM);
			const void *fn = c->nw == 1 ? (const void *)k_sk_scatter_records_staged<1> : (c->nw == 2 ? (const void *)k_sk_scatter_records_staged<2> : (const void *)k_sk_scatter_records_staged<4>);
			SK_CHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
			if (c->nw == 1) hipLaunchKernelGGL(k_sk_scatter_records_staged<1>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_sk_scatter_records_staged<2>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
			else hipLaunchKernelGGL(k_sk_scatter_records_staged<4>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		} else {
		static const size_t l2pad = (size_t)(getenv("SDT_SK_L2_PAD_KB") ? atoi(getenv("SDT_SK_L2_PAD_KB")) : SK_L2_LDS_PAD_KB) * 1024;
		const void *l2fn = c->nw == 1 ? (const void *)k_sk_scatter_records<1> : (c->nw == 2 ? (const void *)k_sk_scatter_records<2> : (const void *)k_sk_scatter_records<4>);
		SK_CHK(hipFuncSetAttribute(l2fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2pad));
		if (c->nw == 1) hipLaunchKernelGGL(k_sk_scatter_records<1>, dim3(nitems), dim3(SK_L2_TPB), l2pad, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_sk_scatter_records<2>, dim3(nitems), dim3(SK_L2_TPB), l2pad, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		else hipLaunchKernelGGL(k_sk_scatter_records<4>, dim3(nitems), dim3(SK_L2_TPB), l2pad, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		}
		SK_CHK(hipGetLastError());
	}
	if (after_l2)
		SK_CHK(hipEventRecord(after_l2, c->stream));
	hipLaunchKernelGGL(k_sk_scan, dim3(1), dim3(1024), 0, c->stream, k.cnt2, k.off2, k.fill2, (int)SK_NBF, (const unsigned long long *)k.kmers2, k.kpre2);
	hipLaunchKernelGGL(k_sk_chunk_place, dim3(g), dim3(256), 0, c->stream, k.p2, k.off2, k.fill2, k.list2);
	SK_CHK(hipGetLastError());
	SK_CHK(hipMemcpyAsync(k.h_kpre2, k.kpre2, (SK_NBF + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	SK_CHK(hipMemcpyAsync(k.h_off2, k.off2, (SK_NBF + 1) * 4, hipMemcpyDeviceToHost, c->stream));
	return SDT_OK;
}

// count pool 2 bucket by bucket (host sync first: the chunk lists of sk_split come back)
static int sk_count_all(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	SK_CHK(hipStreamSynchronize(c->stream));
	k.st_chunks2 = k.h_off2[SK_NBF];
	k.st_flushes++;
	k.stream_flushes++;
	k.l2_in_total += k.h_kpre2[SK_NBF];
	// work items = pieces of buckets of at most SK_COUNT_ITEM_CHUNKS chunks; launches of at most SK_COUNT_KMERS
	// k-mers (every one might be a new node: ensure_room)
	int rc = SDT_OK;
	uint32_t nci = 0;
	SK_CHK(hipMemsetAsync(k.next_item, 0, SK_MAX_COUNT_LAUNCHES * sizeof(uint32_t), c->stream));
	// A launch must find room for every node it may create.  "Every occurrence is a new node" is hopeless for a batch of
	// 2^33 k-mers, so: a first launch of at most 2^26 k-mers under that bound, then launches bounded by twice the rate of
	// new nodes per occurrence seen so far (later data usually brings fewer new nodes, not more; should it bring more, the load
	// factor suffers until the next look but the table cannot fill: see the 95 % rule below).
	// (the items and launches are a pure function of the chunk lists: sdt_count_plan.h, tested on the CPU)
	std::vector<uint32_t> first_item(SK_MAX_COUNT_LAUNCHES + 2);   // first item of every launch
	std::vector<uint64_t> launch_kmers(SK_MAX_COUNT_LAUNCHES + 2);
	uint32_t nlaunches = 0;
	
	if (!sk_plan_count_items(k.h_off2, (const uint64_t *)k.h_kpre2, (uint32_t)SK_NBF, c->kmers_known == 0 ? (1ULL << 26) : SK_COUNT_KMERS, SK_COUNT_KMERS,
	                         SK_MAX_COUNT_LAUNCHES, SK_COUNT_PACK_CHUNKS, SK_COUNT_ITEM_CHUNKS, (uint32_t *)k.h_citems, k.citems_cap,
	                         first_item.data(), launch_kmers.data(), (uint32_t)first_item.size(), &nci, &nlaunches))
		return fail(SDT_ESTATE, "count stage: work item table overflow");
	first_item.resize(nlaunches + 1);
	launch_kmers.resize(nlaunches);
	std::vector<uint32_t> sort_tmp;
	auto guess_of = [&](uint64_t kmers) -> uint64_t {
		uint64_t bound = kmers;
		if (c->kmers_known) {
			const double rate = (double)c->distinct_known / (double)c->kmers_known;
			const uint64_t guess = (uint64_t)((double)kmers * (2.0 * rate < 1.0 ? 2.0 * rate : 1.0)) + (1ULL << 22);
			if (guess < bound) bound = guess;
		}
		return bound;
	};
	const size_t nl = first_item.size() - 1;
	for (size_t l = 0; l < nl && rc == SDT_OK;) {
		const uint32_t i0 = first_item[l];
		if (i0 == first_item[l + 1]) {
			l++;
			continue;
		}
		if (c->kmers_known == 0 && l > 0) {
			rc = sync_stats(c);                      // the first launch has run: its rate of new nodes bounds the rest
			if (rc != SDT_OK) break;
		}
		uint64_t bound = guess_of(launch_kmers[l]), hard = launch_kmers[l];
		rc = ensure_room(c, bound);
		// the guess keeps the load factor; this keeps the table from FILLING should the guess be wrong: whatever the data,
		// the nodes known + every k-mer launched since + this launch must fit 95 % of the slots
		if (rc == SDT_OK && (double)(c->distinct_known + c->hard_since_sync + launch_kmers[l]) > 0.95 * (double)c->slots) {
			rc = sync_stats(c);
			if (rc == SDT_OK && (double)(c->distinct_known + launch_kmers[l]) > 0.95 * (double)c->slots)
				rc = grow_table(c, c->distinct_known + launch_kmers[l]);
		}
		// The planned launches behind this one join it as long as neither rule would have to look at the device's counters for
		// them: a launch boundary is a drained GPU (every workgroup waits for the slowest), and it is only needed where the host
		// decides about the table.  (45 planned launches per step of the 200 M-read workload become about a dozen.)
		size_t m = l;
		while (rc == SDT_OK && c->kmers_known && m + 1 < nl) {
			const uint64_t b2 = guess_of(launch_kmers[m + 1]);
			if ((double)(c->distinct_known + c->kmers_since_sync + bound + b2) > (double)c->slots * MAX_LOAD)
				break;
			if ((double)(c->distinct_known + c->hard_since_sync + hard + launch_kmers[m + 1]) > 0.95 * (double)c->slots)
				break;
			bound += b2;
			hard += launch_kmers[m + 1];
			m++;
		}
		const uint32_t i1 = first_item[m + 1];
		// (largest first over everything this launch hands out -- the plan did it per planned launch; the sort is stable, so the
		// concatenation of sorted runs comes out as one)
		if (m > l)
			sk_plan_largest_first((uint32_t *)k.h_citems, i0, i1, sort_tmp);
		if (rc == SDT_OK)
			SK_CHK(hipMemcpyAsync(k.citems + i0, k.h_citems + i0, (size_t)(i1 - i0) * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
		if (rc == SDT_OK)
			rc = c->nw == 1 ? sk_launch_count<1>(c, i0, i1, (uint32_t)l) : c->nw == 2 ? sk_launch_count<2>(c, i0, i1, (uint32_t)l) : sk_launch_count<4>(c, i0, i1, (uint32_t)l);
		if (rc == SDT_OK) {                              // (only what was launched counts)
			c->kmers_since_sync += bound;
			c->hard_since_sync += hard;
		}
		l = m + 1;
	}
	// (the pinned item list must outlive its copy: the next flush rewrites it only after this stream has drained)
	return rc;
}

// work items of level 2 over the chunks [lo, hi) of bucket b in a list: at most SK_ITEM_CHUNKS chunks each
static int sk_add_items(sdt_ctx::SkState &k, uint32_t &nitems, uint32_t b, uint32_t lo, uint32_t hi)
{
	for (uint32_t c0 = lo; c0 < hi; c0 += SK_ITEM_CHUNKS) {
		if (nitems >= k.items_cap)
			return fail(SDT_EHIP, "super-k-mer pipeline: item table overflow");
		k.h_items[nitems++] = SkItem{b, c0, c0 + SK_ITEM_CHUNKS < hi ? c0 + SK_ITEM_CHUNKS : hi, 0};
	}
	return SDT_OK;
}

static int sk_flush_sharded(sdt_ctx *c);

// everything scattered so far goes into the node table: seal + list the level-1 chunks, split every level-1 bucket,
// list the level-2 chunks, count bucket by bucket
static int sk_flush(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	if (!k.ready || k.pending_kmers == 0 || k.flushing)
		return SDT_OK;
	if (c->comm.nranks > 1)
		return fail(SDT_ESTATE, "sharded context: the pipeline is drained by the collective calls (sdt_gpu_count_reads_sharded)");
	k.flushing = true;
	EventPair *ev = next_event(c), *ev2 = next_event(c);
	if (!ev || !ev2) { k.flushing = false; return fail(SDT_EHIP, "hipEventCreate failed"); }
	ev = ev2 - 1;                                    // next_event may have moved the vector
	ev->kmers = ev2->kmers = 0;
	ev->stage = SDT_STAGE_SK_SPLIT;
	ev2->stage = SDT_STAGE_SK_COUNT;
	int rc = hipEventRecord(ev->a, c->stream) == hipSuccess ? SDT_OK : fail(SDT_EHIP, "hipEventRecord failed");
	if (rc == SDT_OK) rc = sk_list1(c);
	uint32_t nitems = 0;
	for (uint32_t b = 0; b < (uint32_t)SK_NB1 && rc == SDT_OK; b++)
		rc = sk_add_items(k, nitems, b, k.h_off1[b], k.h_off1[b + 1]);
	if (rc == SDT_OK) rc = sk_split(c, k.p1, k.list1, nitems, nullptr);
	// pool 1 is free again: the next batch may scatter while this one is counted (same stream: in order)
	if (rc == SDT_OK) rc = sk_reset_pool1(c);
	if (rc == SDT_OK && (hipEventRecord(ev->b, c->stream) != hipSuccess || hipEventRecord(ev2->a, c->stream) != hipSuccess))
		rc = fail(SDT_EHIP, "hipEventRecord failed");
	if (rc == SDT_OK) {
		k.pending_kmers = 0;
		rc = sk_count_all(c);
	}
	if (hipEventRecord(ev2->b, c->stream) != hipSuccess && rc == SDT_OK)
		rc = fail(SDT_EHIP, "hipEventRecord failed");
	k.flushing = false;
	return rc;
}

template <int NW> static Table<NW> sk_tbl(const sdt_ctx *c, bool allow_direct)
{
	Table<NW> t = flat_of<NW>(c);
	if (!allow_direct)
		t.ent = nullptr;
	return t;
}

// one launch of the level-1 scatter over reads [0, nr) of a device-resident batch (ordinals from `ob`)
// (allow_direct = false: a record that finds no chunk is an error, not a put into the local table -- sharded contexts, where
// the local table owns only some buckets, and the range-weighing sample, whose reads are scattered a second time)
static int sk_scatter_launch(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nr, uint64_t max_read_len, uint64_t ob,
                             bool allow_direct = true)
{
	sdt_ctx::SkState &k = c->sk;
	const uint64_t per_read = max_read_len - c->K + 1;
	const SkGeo geo = sk_geo(c->K, max_read_len);
	const int m = sk_minimizer_len(c->K), ncap = sk_max_run(c->K, c->nw);
	const uint64_t ntiles = (nr + SK_TILE_READS - 1) / SK_TILE_READS;
	const unsigned grid = (unsigned)(ntiles < k.wgs ? ntiles : k.wgs);
	EventPair *ev = next_event(c);
	if (!ev)
		return fail(SDT_EHIP, "hipEventCreate failed");
	ev->kmers = nr * per_read;
	ev->stage = SDT_STAGE_SK_SCATTER;
	HIPCHK(hipEventRecord(ev->a, c->stream));
#define SK_SCATTER(NW)                                                                                                             \
	do {                                                                                                                       \
		HIPCHK(hipFuncSetAttribute((const void *)k_sk_scatter_reads<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.smem)); \
		hipLaunchKernelGGL(k_sk_scatter_reads<NW>, dim3(grid), dim3(TPB), geo.smem, c->stream, d_words, d_offs, nr, c->K, m, ncap, \
		                   geo.mtw, geo.tile_words, geo.hv_words, geo.hv2_words, geo.bits_words, k.p1, k.cursors, k.blk, k.cnt1, sk_tbl<NW>(c, allow_direct), c->d_stats, ob, c->ord_stride); \
	} while (0)
	// one lane per read where the window length has an instantiation and the run list can hold a read
	const int w = c->K - m + 1;
	static const bool no_seq = getenv("SDT_SK_STRIPS") != NULL;          // A/B switch (tools/)
	// (instantiated: every odd window of 1-word keys with K >= 17 and of 2-word keys, i.e. every odd K from 17 to 63)
	const bool seq1 = c->nw == 1 && (w & 1) && w >= 9 && w <= 21 && per_read <= (uint64_t)SK_SEQ_MAX_KMERS;
	const bool seq2 = c->nw == 2 && (w & 1) && w >= 23 && w <= 53 && per_read <= (uint64_t)SK_SEQ_MAX_KMERS;
	if ((seq1 || seq2) && !no_seq) {
		SkSeqLaunch a;
		a.words = d_words; a.offs = d_offs; a.nreads = nr; a.K = c->K; a.m = m; a.ncap = ncap;
		a.mtw = (int)(((uint64_t)SK_SEQ_TILE * max_read_len + 16 + 15) / 16) + TAIL_PAD + 1;
		a.pool = k.p1; a.cursors = k.cursors; a.blk = k.blk; a.cnt = k.cnt1; a.stats = c->d_stats;
		a.ord_base = ob; a.ord_stride = c->ord_stride; a.max_wgs = k.wgs; a.cu_count = c->cu_count; a.stream = c->stream;
		HIPCHK(seq1 ? sk_seq_launch_nw1(w, a, sk_tbl<1>(c, allow_direct)) : (w <= 33 ? sk_seq_launch_nw2_lo(w, a, sk_tbl<2>(c, allow_direct))
		            : (w <= 43 ? sk_seq_launch_nw2_mid(w, a, sk_tbl<2>(c, allow_direct)) : sk_seq_launch_nw2_hi(w, a, sk_tbl<2>(c, allow_direct)))));
	} else if (c->nw == 1) SK_SCATTER(1);
	else if (c->nw == 2) SK_SCATTER(2);
	else SK_SCATTER(4);
#undef SK_SCATTER
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(ev->b, c->stream));
	k.pending_kmers += nr * per_read;
	return SDT_OK;
}

// k-mers a batch may hold before it is flushed: the pools' capacity -- except for the first batches of a stream whose length
// the caller has announced (sdt_gpu_hint_total_kmers): 1/16 of the job, then 1/8, 1/4 ... .  The kernels need about twice the
// time of the copies, so a batch's copies hide behind the counting of the batches before it as long as it is at most about
// twice their size; one large batch after a small first one left the GPU waiting for 9 GB of copies (measured: 66 instead
// of 68 G k-mers/s from host memory), equal quarters of the job merge more often than they must.
static uint64_t sk_batch_limit(const sdt_ctx *c)
{
	const sdt_ctx::SkState &k = c->sk;
	if (c->expect_kmers && c->expect_kmers / 16 >= (1ULL << 27) && k.stream_flushes < 4) {
		const uint64_t lim = (c->expect_kmers / 16) << k.stream_flushes;
		if (lim < k.cap_kmers)
			return lim;
	}
	return k.cap_kmers;
}

// chop + scatter a device-resident batch into the level-1 buckets (flushing whenever the pools are full)
static int sk_scatter(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads, uint64_t max_read_len)
{
	sdt_ctx::SkState &k = c->sk;
	const uint64_t per_read = max_read_len - c->K + 1;
	int rc = SDT_OK;
	if (k.ready && k.pending_kmers && !k.cap_is_max && k.cap_kmers < k.pending_kmers + nreads * per_read && k.cap_kmers < (1ULL << 31))
		rc = sk_flush(c);                            // the pools are about to be replaced by larger ones
	if (rc == SDT_OK) {
		// without SDT_FLAG_PARTITION the pipeline only runs for jobs past 2^27 k-mers: start with pools for 2^31 at once
		// (sk_alloc halves that until it fits the free memory) instead of growing there batch by batch
		uint64_t want = k.pending_kmers + nreads * per_read;
		if (!(c->flags & SDT_FLAG_PARTITION) && want < (1ULL << 31))
			want = 1ULL << 31;
		// a caller that streams its reads in and has said how much is coming (sdt_gpu_hint_total_kmers): pools for 2^32 k-mers from
		// the first batch on (26 GiB at K = 31), not grown there batch by batch.  NOT pools for the whole job any more: fewer, larger
		// batches merge a little less (quarters of a 14 G k-mer job measured 3 % slower than one batch), but the 103 GiB of pools of
		// that job cost 1.4 - 4.7 s to allocate whenever the box's memory had been used before (sdt_mem.hip) -- a hundred times the gain.
		if (c->expect_kmers > want) {
			const uint64_t lim = 1ULL << clamp_int(env_int("SDT_SK_HINT_BATCH_LOG2", 32), 24, 34);
			const uint64_t hinted = c->expect_kmers < lim ? c->expect_kmers : lim;
			if (hinted > want) want = hinted;
		}
		rc = sk_alloc(c, want, per_read);
	}
	if (rc != SDT_OK)
		return rc;
	for (uint64_t r0 = 0; r0 < nreads;) {
		if (k.pending_kmers + per_read * SK_TILE_READS > sk_batch_limit(c)) {
			rc = sk_flush(c);
			// a stream that keeps filling SMALL pools gets larger ones: fewer batches = fewer merges per distinct key.  Past 2^31
			// k-mers they stay: replacing tens of GiB was seen to stall for seconds in hipFree / hipMalloc now and then.
			if (rc == SDT_OK && !k.cap_is_max && k.cap_kmers < (1ULL << 31))
				rc = sk_alloc(c, k.cap_kmers * 2, per_read);
			if (rc != SDT_OK)
				return rc;
		}
		uint64_t nr = (sk_batch_limit(c) - k.pending_kmers) / per_read / SK_TILE_READS * SK_TILE_READS;
		if (nr > nreads - r0) nr = nreads - r0;
		rc = sk_scatter_launch(c, d_words, d_offs + r0, nr, max_read_len, c->ord_base + r0 * c->ord_stride);
		if (rc != SDT_OK)
			return rc;
		r0 += nr;
	}
	return SDT_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: ranks own contiguous ranges of the level-1 buckets; level-1 chunks travel to their owner
// ------------------------------------------------------------------------------------------------
static void shard_free(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	for (int i = 0; i < 2; i++) {
		if (h.send[i]) (void)hipFree(h.send[i]);
		if (h.recv[i]) (void)hipFree(h.recv[i]);
		if (h.send_meta[i]) (void)hipFree(h.send_meta[i]);
		if (h.recv_meta[i]) (void)hipFree(h.recv_meta[i]);
		if (h.ev_gather[i]) (void)hipEventDestroy(h.ev_gather[i]);
		if (h.ev_xdone[i]) (void)hipEventDestroy(h.ev_xdone[i]);
		if (h.ev_l2[i]) (void)hipEventDestroy(h.ev_l2[i]);
	}
	if (h.iota) (void)hipFree(h.iota);
	h = sdt_ctx::Shard();
}

static int shard_alloc(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	if (h.send[0] && h.send_chunks >= k.p1.chunks)
		return SDT_OK;
	HIPCHK(hipStreamSynchronize(c->stream));
	shard_free(c);
	const size_t cw = (size_t)SK_CAP1 * sk_rec_words(c->nw) * 8;
	h.send_chunks = k.p1.chunks;
	h.recv_chunks = k.p1.chunks + k.p1.chunks / 4;   // a rank receives ~ what it sends; head room for unequal buckets
	if (getenv("SDT_SHARD_RECV_CHUNKS"))             // (tests: force sub-rounds)
		h.recv_chunks = (uint32_t)strtoul(getenv("SDT_SHARD_RECV_CHUNKS"), nullptr, 10);
	for (int i = 0; i < 2; i++) {
		HIPCHK(hipMalloc((void **)&h.send[i], (size_t)h.send_chunks * cw));
		HIPCHK(hipMalloc((void **)&h.recv[i], (size_t)h.recv_chunks * cw));
		HIPCHK(hipMalloc((void **)&h.send_meta[i], (size_t)h.send_chunks * 4));
		HIPCHK(hipMalloc((void **)&h.recv_meta[i], (size_t)h.recv_chunks * 4));
		HIPCHK(hipEventCreateWithFlags(&h.ev_gather[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h.ev_xdone[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h.ev_l2[i], hipEventDisableTiming));
	}
	HIPCHK(hipMalloc((void **)&h.iota, (size_t)h.recv_chunks * 4));
	hipLaunchKernelGGL(k_sk_iota, dim3(1024), dim3(256), 0, c->stream, h.iota, h.recv_chunks);
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

// split + count what the last exchange delivered
static int shard_finish_pending(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	if (!h.pending)
		return SDT_OK;
	h.pending = false;
	const int slot = h.pending_slot;
	EventPair *ev = next_event(c), *ev2 = next_event(c);
	if (!ev || !ev2) return fail(SDT_EHIP, "hipEventCreate failed");
	ev = ev2 - 1;
	ev->kmers = ev2->kmers = 0;
	ev->stage = SDT_STAGE_SK_SPLIT;
	ev2->stage = SDT_STAGE_SK_COUNT;
	HIPCHK(hipStreamWaitEvent(c->stream, h.ev_xdone[slot], 0));
	HIPCHK(hipEventRecord(ev->a, c->stream));
	if (h.items.size() > k.items_cap)
		return fail(SDT_EHIP, "super-k-mer pipeline: item table overflow");
	memcpy(k.h_items, h.items.data(), h.items.size() * sizeof(SkItem));
	SkPool src = {h.recv[slot], h.recv_meta[slot], nullptr, h.recv_chunks};
	int rc = sk_split(c, src, h.iota, (uint32_t)h.items.size(), h.ev_l2[slot]);
	if (rc != SDT_OK) return rc;
	h.l2_recorded[slot] = true;                      // (only an event that was recorded may be waited for)
	HIPCHK(hipEventRecord(ev->b, c->stream));
	HIPCHK(hipEventRecord(ev2->a, c->stream));
	rc = sk_count_all(c);
	HIPCHK(hipEventRecord(ev2->b, c->stream));
	return rc;
}

// COLLECTIVE.  Level-1 chunks scattered since the last call go to the ranks that own their buckets; what the previous
// call's exchange delivered is split and counted meanwhile.  Sub-rounds when a rank would receive more than its buffer holds.
static int sk_flush_sharded(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	Comm &cm = c->comm;
	const int n = cm.nranks, me = cm.rank;
	k.exchanged = true;
	int rc = sk_list1(c);
	if (rc != SDT_OK) return rc;
	// everybody's chunk counts per bucket
	std::vector<uint32_t> mat((size_t)n * (SK_NB1 + 1));
	rc = cm.allgather_host(k.h_off1, mat.data(), (SK_NB1 + 1) * sizeof(uint32_t));
	if (rc != SDT_OK) return rc;
	auto M = [&](int r, uint32_t b) { return shard_mat(mat.data(), r, b); };
	const uint32_t *blo = h.ranges;
	// sub-rounds, pieces and buffer layouts: pure functions of the matrix (sdt_shard_plan.h) -- every rank computes every
	// rank's layout from it, so all agree without another message
	const uint32_t S = shard_subrounds(mat.data(), n, blo, h.recv_chunks);
	const size_t cw = (size_t)SK_CAP1 * sk_rec_words(c->nw) * 8;
	for (uint32_t t = 0; t < S; t++) {
		const int slot = (int)(h.round & 1);
		ShardRound sr;
		shard_round(mat.data(), n, me, blo, t, S, sr);
		auto piece = [&](int s2, int d, uint32_t &lo, uint32_t &hi) { shard_piece(mat.data(), blo, s2, d, t, S, lo, hi); };
		SkGatherPlan plan;
		memset(&plan, 0, sizeof plan);
		plan.n = n;
		plan.self = me;
		std::vector<void *> sp(n), rp(n), smp(n), rmp(n);
		std::vector<size_t> sb(n, 0), rb(n, 0), smb(n, 0), rmb(n, 0), oboff((size_t)n * n, 0), obmoff((size_t)n * n, 0);
		const uint32_t send_at = sr.send_total, recv_at = sr.recv_total;
		std::vector<SkItem> cur;                     // level-2 work items of THIS exchange (h.items still describes the last one)
		for (int p = 0; p < n; p++) {
			plan.begin[p] = sr.send_begin[p];
			plan.pre[p + 1] = plan.pre[p] + sr.send_count[p];
			plan.dst0[p] = sr.send_at[p];
			if (p != me) {
				sp[p] = (uint8_t *)h.send[slot] + (size_t)sr.send_at[p] * cw;
				smp[p] = h.send_meta[slot] + sr.send_at[p];
				sb[p] = (size_t)sr.send_count[p] * cw;
				smb[p] = (size_t)sr.send_count[p] * 4;
			}
		}
		for (int s2 = 0; s2 < n; s2++) {             // receive buffer: one run per source, rank order (mine included)
			uint32_t lo, hi;
			piece(s2, me, lo, hi);
			const uint32_t at = sr.recv_at[s2];
			rp[s2] = (uint8_t *)h.recv[slot] + (size_t)at * cw;
			rmp[s2] = h.recv_meta[slot] + at;
			rb[s2] = (size_t)(hi - lo) * cw;
			rmb[s2] = (size_t)(hi - lo) * 4;
			// level-2 work items over this run: its chunks are in bucket order
			for (uint32_t b = blo[me]; b < blo[me + 1] && rc == SDT_OK; b++) {
				const uint32_t x0 = M(s2, b) > lo ? M(s2, b) : lo, x1 = M(s2, b + 1) < hi ? M(s2, b + 1) : hi;
				for (uint32_t c0 = at + (x0 - lo); x1 > x0 && c0 < at + (x1 - lo); c0 += SK_ITEM_CHUNKS) {
					const uint32_t c1 = c0 + SK_ITEM_CHUNKS < at + (x1 - lo) ? c0 + SK_ITEM_CHUNKS : at + (x1 - lo);
					cur.push_back(SkItem{b, c0, c1, 0});
				}
			}
		}
		if (send_at > h.send_chunks || recv_at > h.recv_chunks)
			return fail(SDT_EFULL, "exchange buffers too small: %u / %u chunks to send, %u / %u to receive", send_at, h.send_chunks, recv_at, h.recv_chunks);
		// outbox layout of every rank (shared-memory transport): destinations in rank order
		if (cm.kind == 2)
			for (int s2 = 0; s2 < n; s2++) {
				size_t at = 0, mat_at = 0;
				for (int d = 0; d < n; d++) {
					if (d == s2) continue;
					uint32_t lo, hi;
					piece(s2, d, lo, hi);
					oboff[(size_t)s2 * n + d] = at;
					at += (size_t)(hi - lo) * cw;
				}
				for (int d = 0; d < n; d++) {
					if (d == s2) continue;
					uint32_t lo, hi;
					piece(s2, d, lo, hi);
					obmoff[(size_t)s2 * n + d] = at + mat_at;      // metas behind all payloads
					mat_at += (size_t)(hi - lo) * 4;
				}
			}
		// G: the send buffer of this slot must have left (exchange of two rounds ago)
		if (h.x_recorded[slot])
			HIPCHK(hipStreamWaitEvent(c->stream, h.ev_xdone[slot], 0));
		if (plan.pre[n]) {
			const unsigned g = (unsigned)c->cu_count * 8;
			if (c->nw == 1) hipLaunchKernelGGL(k_sk_gather<SkFmt<1>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			else if (c->nw == 2) hipLaunchKernelGGL(k_sk_gather<SkFmt<2>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			else hipLaunchKernelGGL(k_sk_gather<SkFmt<4>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			HIPCHK(hipGetLastError());
		}
		HIPCHK(hipEventRecord(h.ev_gather[slot], c->stream));
		if (t + 1 == S) {                            // pool 1 is free: the next round may scatter while this one travels
			rc = sk_reset_pool1(c);
			if (rc != SDT_OK) return rc;
		}
		// X: on the exchange stream, after the gather and after level 2 has drained this slot's receive buffer
		HIPCHK(hipStreamWaitEvent(cm.xstream, h.ev_gather[slot], 0));
		if (h.l2_recorded[slot])
			HIPCHK(hipStreamWaitEvent(cm.xstream, h.ev_l2[slot], 0));
		{
			// payloads and metas in ONE grouped exchange (one event pair: the time sdt_gpu_comm_stats reports covers both)
			void *const *const sps[2] = {sp.data(), smp.data()}, *const *const rps[2] = {rp.data(), rmp.data()};
			const size_t *const sbs[2] = {sb.data(), smb.data()}, *const rbs[2] = {rb.data(), rmb.data()}, *const obs[2] = {oboff.data(), obmoff.data()};
			rc = cm.exchange(2, sps, sbs, rps, rbs, obs);
		}
		if (rc != SDT_OK) return rc;
		HIPCHK(hipEventRecord(h.ev_xdone[slot], cm.xstream));
		h.x_recorded[slot] = true;
		// B: meanwhile, split + count what the previous exchange brought
		rc = shard_finish_pending(c);
		if (rc != SDT_OK) return rc;
		h.items.swap(cur);
		h.pending = true;
		h.pending_slot = slot;
		h.round++;
	}
	return SDT_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------

template <int NW>
static int build_patch_table(sdt_ctx *c, const uint64_t *pkeys, const uint64_t *pinfo, uint64_t np)
{
	uint64_t slots = 1024;
	while (slots < 2 * np + 2)
		slots <<= 1;
	std::vector<PatchEnt<NW>> tab(slots);
	for (auto &e : tab) {
		for (int w = 0; w < NW; w++) e.key[w] = KEY_EMPTY;
		e.info = 0;
	}
	for (uint64_t i = 0; i < np; i++) {
		Key<NW> k;
		for (int w = 0; w < NW; w++) k.w[w] = pkeys[i * NW + w];
		uint64_t s = key_hash<NW>(k) & (slots - 1);
		while (tab[s].key[0] != KEY_EMPTY) s = (s + 1) & (slots - 1);
		for (int w = 0; w < NW; w++) tab[s].key[w] = k.w[w];
		tab[s].info = pinfo[i];
	}
	if (c->d_patch) HIPCHK(hipFree(c->d_patch));
	c->d_patch = nullptr;
	HIPCHK(hipMalloc(&c->d_patch, slots * sizeof(PatchEnt<NW>)));
	HIPCHK(hipMemcpy(c->d_patch, tab.data(), slots * sizeof(PatchEnt<NW>), hipMemcpyHostToDevice));
	c->patch_slots = slots;
	return SDT_OK;
}


extern "C" {

const char *sdt_gpu_last_error(void) { return g_err; }
int sdt_gpu_abi_version(void) { return SDT_ABI_VERSION; }

uint64_t sdt_owner_hash(const uint64_t *key_words_msw_first, int nwords)
{
	if (nwords == 1) { Key<1> k{{key_words_msw_first[0]}}; return key_hash<1>(k); }
	if (nwords == 2) { Key<2> k{{key_words_msw_first[0], key_words_msw_first[1]}}; return key_hash<2>(k); }
	Key<4> k{{key_words_msw_first[0], key_words_msw_first[1], key_words_msw_first[2], key_words_msw_first[3]}};
	return key_hash<4>(k);
}

int sdt_gpu_init(sdt_ctx **out, int device, int K, uint64_t est_distinct, uint32_t flags)
{
	if (!out)
		return fail(SDT_EINVAL, "ctx is NULL");
	*out = nullptr;
	if (K < 13 || K > 127 || (K & 1) == 0)
		return fail(SDT_EINVAL, "K must be odd and in 13..127 (got %d); apply call_pregraph's clamp first", K);
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(SDT_ENODEV, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	if (device < 0 || device >= ndev)
		return fail(SDT_EINVAL, "device %d out of range (have %d)", device, ndev);
	{   // SDT_SYNC=block|yield|spin: how host threads wait for the device (a host that parses on every core it may use wants its
		// waiting thread off the CPU; the default is the runtime's own choice).  Must be set before the device is first used.
		const char *sm = getenv("SDT_SYNC");
		if (sm && *sm) {
			const unsigned fl = !strcmp(sm, "block") ? hipDeviceScheduleBlockingSync : !strcmp(sm, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleSpin;
			(void)hipSetDeviceFlags(fl);
			(void)hipGetLastError();
		}
	}
	e = hipSetDevice(device);
	if (e != hipSuccess)
		return fail(SDT_ENODEV, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess)
		return fail(SDT_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(SDT_ENODEV, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

	sdt_ctx *c = new (std::nothrow) sdt_ctx();
	if (!c)
		return fail(SDT_ENOMEM, "out of host memory");
	g_live_ctx.fetch_add(1);
	c->device = device;
	c->K = K;
	c->nw = K <= 31 ? 1 : (K <= 63 ? 2 : 4);
	c->flags = (flags & SDT_FLAG_CONTIG_INDEX) ? (flags | SDT_FLAG_TRACK_FIRST) : flags;   // the index lives in the first-occurrence slot
	c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (est_distinct == 0)
		est_distinct = 1ULL << 22;
	// one node table for both kernel families (the direct one counts into it with one atomic per occurrence, the locality pipeline
	// merges its LDS tables into it), sized by the caller's estimate and grown by k_rehash.  (Round 5 also built a node LOG folded
	// into a bucket-major table; it ran at the speed of these merges and was removed in round 6: DESIGN.md section 3.)
	c->slots = flat_slots_for(est_distinct);
#define INIT_CHK(expr)                                                                    \
	do {                                                                                  \
		hipError_t e2_ = (expr);                                                          \
		if (e2_ != hipSuccess) {                                                          \
			int rc_ = fail(e2_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e2_)); \
			sdt_gpu_destroy(c);                                                           \
			return rc_;                                                                   \
		}                                                                                 \
	} while (0)
	INIT_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	INIT_CHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
	for (int i = 0; i < sdt_ctx::NSTAGE; i++) {
		INIT_CHK(hipEventCreateWithFlags(&c->buf_free[i], hipEventDisableTiming));
		INIT_CHK(hipEventCreateWithFlags(&c->copied[i], hipEventDisableTiming));
	}
	INIT_CHK(hipMalloc((void **)&c->d_stats, sizeof(Stats)));
	INIT_CHK(hipHostMalloc((void **)&c->h_stats, sizeof(Stats), hipHostMallocDefault));
	INIT_CHK(hipMalloc((void **)&c->d_hist, 257 * sizeof(unsigned long long)));
	int rc = alloc_table(c, c->slots, &c->d_ent, &c->d_aux, &c->d_first);
	if (rc != SDT_OK) {
		sdt_gpu_destroy(c);
		return rc;
	}
	rc = sdt_gpu_reset(c);
	if (rc != SDT_OK) {
		sdt_gpu_destroy(c);
		return rc;
	}
	*out = c;
	return SDT_OK;
}

int sdt_gpu_destroy(sdt_ctx *c)
{
	if (!c)
		return SDT_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
	for (auto &p : c->ev) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
	for (int i = 0; i < sdt_ctx::NSTAGE; i++) {
		if (c->d_words[i]) (void)hipFree(c->d_words[i]);
		if (c->d_offs[i]) (void)hipFree(c->d_offs[i]);
		if (c->buf_free[i]) (void)hipEventDestroy(c->buf_free[i]);
		if (c->copied[i]) (void)hipEventDestroy(c->copied[i]);
	}
	if (c->d_ent) (void)hipFree(c->d_ent);
	if (c->d_aux) (void)hipFree(c->d_aux);
	if (c->d_first) (void)hipFree(c->d_first);
	if (c->d_stats) (void)hipFree(c->d_stats);
	if (c->h_stats) (void)hipHostFree(c->h_stats);
	if (c->d_hist) (void)hipFree(c->d_hist);
	keep_release(c);
	if (c->d_patch) (void)hipFree(c->d_patch);
	if (c->d_arcs) (void)hipFree(c->d_arcs);
	if (c->d_idx) (void)hipFree(c->d_idx);
	if (c->gx) sdti::graph_ext_free(c->gx);
	if (c->d_ctg_ids) (void)hipFree(c->d_ctg_ids);
	if (c->d_ctg_len) (void)hipFree(c->d_ctg_len);
	if (c->d_ctg_twin) (void)hipFree(c->d_ctg_twin);
	if (c->d_hit_cursor) (void)hipFree(c->d_hit_cursor);
	for (int i = 0; i < 5; i++) if (c->ab[i]) (void)hipFree(c->ab[i]);
	sk_free(c);
	shard_free(c);
	c->comm.close_all();
	if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
	if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
	delete c;
	if (g_live_ctx.fetch_sub(1) == 1) (void)sdti::mem_trim();       // the last context of the process: the arena's memory goes back to the driver
	return SDT_OK;
}

int sdt_gpu_reset(sdt_ctx *c)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	if ((c->flags & SDT_FLAG_TRACK_FIRST) && !c->d_first)           // (dropped once the device had laid the graph out: sdti::drop_first)
		HIPCHK(hipMalloc((void **)&c->d_first, c->slots * sizeof(uint64_t)));
	int rc = launch_clear(c, c->d_ent, c->d_aux, c->d_first, c->slots);
	if (rc != SDT_OK)
		return rc;
	HIPCHK(hipMemsetAsync(c->d_stats, 0, sizeof(Stats), c->stream));
	c->sk.l2_in_total = 0;
	c->sk.stream_flushes = 0;
	c->sk.exchanged = false;
	c->distinct_known = 0;
	c->kmers_known = 0;
	c->kmers_since_sync = 0;
	c->hard_since_sync = 0;
	c->kmers_total_host = 0;
	c->kmers_offered = 0;
	c->expect_kmers = 0;
	c->ord_base = 0;
	c->push_ord_base = 0;
	c->push_ord_stride = 1;
	c->staged.clear();
	c->staged_head = 0;
	c->ord_stride = 1;
	if (c->sk.ready) {                               // records scattered but not counted belong to the run being forgotten
		const int rcr = sk_reset_pool1(c);
		if (rcr != SDT_OK)
			return rcr;
	}
	keep_release(c);
	c->ctg_ord = 0;
	c->index_final = false;
	c->paths_loaded = false;
	c->sh.pending = false;                           // (a sharded call that failed half-way leaves nothing behind)
	c->sh.items.clear();
	if (c->d_idx) (void)hipFree(c->d_idx);
	c->d_idx = nullptr;
	c->idx_slots = c->idx_n = 0;
	return SDT_OK;
}

// pinned host memory for batches that are pushed asynchronously (the DMA engine reads it directly; pageable memory goes through the
// runtime's own staging copy at a fraction of the link).  Any host thread may call these.
void *sdt_gpu_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}
void sdt_gpu_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

int sdt_gpu_key_words(const sdt_ctx *c) { return c ? c->nw : 0; }
uint64_t sdt_gpu_table_slots(const sdt_ctx *c) { return c ? view_slots(c) : 0; }
void *sdt_gpu_stream(const sdt_ctx *c) { return c ? (void *)c->stream : nullptr; }
int sdt_gpu_set_read_ordinal(sdt_ctx *c, uint64_t base, uint64_t stride)
{
	if (!c || stride == 0)
		return fail(SDT_EINVAL, "bad argument");
	c->push_ord_base = base;
	c->push_ord_stride = stride;
	if (c->staged_head == c->staged.size()) {            // nothing queued: the launch cursor follows at once
		c->ord_base = base;
		c->ord_stride = stride;
	}
	return SDT_OK;
}

// run the kernels on a stream the caller owns (e.g. the stream a collective library orders against)
int sdt_gpu_set_stream(sdt_ctx *c, void *hip_stream)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->own_stream) {
		HIPCHK(hipStreamDestroy(c->stream));
		c->own_stream = false;
	}
	if (hip_stream) {
		c->stream = (hipStream_t)hip_stream;
	} else {
		HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->own_stream = true;
	}
	return SDT_OK;
}

// launch the fused chop+insert kernel on a device-resident batch, in chunks of reads small enough that
// "every occurrence is a new node" cannot overflow the table between two looks at the node counter.
static const uint64_t CHUNK_KMERS = 1ULL << 27;



static int launch_count(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads,
                        uint64_t max_read_len)
{
	if (nreads == 0)
		return SDT_OK;
	if (max_read_len < (uint64_t)c->K + 1)
		return SDT_OK;                               // no read can hold a k-mer (prlHashReads.c:592)
	const int mtw = tile_words_for(max_read_len);
	const size_t smem = tile_smem_bytes(mtw);
	if (smem > 64 * 1024)
		return fail(SDT_EINVAL, "max read length %llu needs %zu B of LDS per tile (limit 64 KiB)",
		            (unsigned long long)max_read_len, smem);
	const uint64_t per_read = max_read_len - c->K + 1;
	// The locality pipeline is opt-in in round 1: on MI355X it measures 11 G k-mers/s against the direct
	// kernel's 19.5 G (profiles/r1/partition_pipeline_50M.md has the per-stage rates and what has to change).
	// default: the locality pipeline wherever it applies (2.2x the direct kernel on 200 M x 150 bp, K = 31); SDT_FLAG_DIRECT /
	// SDT_FLAG_PARTITION force one family (the latter still needs a geometry the pipeline can take)
	// (a small job is not worth the pipeline's fixed cost -- two host syncs and scans over 2^18 buckets, ~3 ms -- unless asked for)
	const bool sk_small = !(c->flags & SDT_FLAG_PARTITION) && !c->sk.pending_kmers && c->kmers_offered + nreads * per_read < (1ULL << 27) &&
	                      c->expect_kmers < (1ULL << 27);      // (a caller that knows more is coming says so: sdt_gpu_hint_total_kmers)
	c->kmers_offered += nreads * per_read;
	const bool sk_ord_ok = c->ord_base + nreads * c->ord_stride < SK_MAX_READ_ORDINAL;      // what a record header can number
	if (!(c->flags & SDT_FLAG_DIRECT) && !sk_small && sk_ord_ok && sk_applicable(c, max_read_len)) {
		const int rcs = sk_scatter(c, d_words, d_offs, nreads, max_read_len);
		if (rcs == SDT_OK)
			c->ord_base += nreads * c->ord_stride;     // the next batch continues the read stream
		return rcs;
	}
	uint64_t chunk_reads = CHUNK_KMERS / per_read;
	chunk_reads = chunk_reads / TILE_READS * TILE_READS;
	if (chunk_reads < TILE_READS)
		chunk_reads = TILE_READS;
	for (uint64_t r0 = 0; r0 < nreads; r0 += chunk_reads) {
		const uint64_t nr = nreads - r0 < chunk_reads ? nreads - r0 : chunk_reads;
		const uint64_t upper = nr * per_read;
		int rc = ensure_room(c, upper);
		if (rc != SDT_OK)
			return rc;
		const uint64_t ntiles = (nr + TILE_READS - 1) / TILE_READS;
		uint64_t grid = ntiles;
		const uint64_t cap = (uint64_t)c->cu_count * 8;
		if (grid > cap) grid = cap;
		EventPair *ev = next_event(c);
		if (!ev)
			return fail(SDT_EHIP, "hipEventCreate failed");
		ev->kmers = upper;
		HIPCHK(hipEventRecord(ev->a, c->stream));
		// offsets are absolute base indices into d_words, so a sub-range of reads is just a shifted pointer
		if (c->nw == 1)
			hipLaunchKernelGGL(k_count_reads<1>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<1>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		else if (c->nw == 2)
			hipLaunchKernelGGL(k_count_reads<2>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<2>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		else
			hipLaunchKernelGGL(k_count_reads<4>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<4>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(ev->b, c->stream));
		c->kmers_since_sync += upper;
	}
	c->ord_base += nreads * c->ord_stride;         // the next batch continues the read stream
	return SDT_OK;
}

// enqueue one host batch: H2D on the copy stream into the next buffer of the ring, kernels behind it.  Returns without waiting
// for the copy; *ticket (may be NULL) names it for sdt_gpu_push_wait.
// offsets of a batch of equal-length reads, made where they are used (8 B per read that need not cross PCIe: 21 % of a 150-bp batch)
__global__ __launch_bounds__(256) void k_fixed_offsets(uint64_t *offs, uint64_t nreads, uint64_t len)
{
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= nreads; i += (uint64_t)gridDim.x * 256ull)
		offs[i] = i * len;
}

static int push_reads_enqueue(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                              uint64_t *ticket, uint64_t fixed_len = 0)
{
	if (!c || (!packed_words && nwords) || (!offsets && !fixed_len))
		return fail(SDT_EINVAL, "NULL argument");
	if (ticket) *ticket = c->push_ticket;
	if (nreads == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	// nothing queued: the launch cursor is the truth (calls that count device-resident reads advance it on their own, and
	// sdt_gpu_set_read_ordinal sets both when the queue is empty)
	if (c->staged_head == c->staged.size()) {
		c->push_ord_base = c->ord_base;
		c->push_ord_stride = c->ord_stride;
	}
	// batch geometry from the host copy of the offsets
	uint64_t kmers = 0, maxlen = 0, bad = 0;
	const uint64_t Kp1 = (uint64_t)c->K + 1;
	if (fixed_len) {
		maxlen = fixed_len;
		kmers = fixed_len >= Kp1 ? nreads * (fixed_len - Kp1 + 2) : 0;
	} else {
		for (uint64_t i = 0; i < nreads; i++) {          // (branch-free: one pass over a million offsets per batch, vectorised)
			const uint64_t len = offsets[i + 1] - offsets[i];
			bad |= len >> 63;                            // offsets[i + 1] < offsets[i]
			maxlen = len > maxlen ? len : maxlen;
			kmers += len >= Kp1 ? len - Kp1 + 2 : 0;
		}
	}
	if (bad)
		return fail(SDT_EINVAL, "offsets not monotonic");
	const uint64_t total_bases = fixed_len ? nreads * fixed_len : offsets[nreads];
	if (((total_bases + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words, got %llu",
		            (unsigned long long)(((total_bases + 15) >> 4) + TAIL_PAD), TAIL_PAD, (unsigned long long)nwords);
	const int b = c->next_buf;
	uint32_t *dw;
	uint64_t *dof;
	if (c->flags & SDT_FLAG_KEEP_READS) {
		// the batch stays resident for the second pass: own buffers instead of the recycled staging ring
		sdt_ctx::KeptBatch kb;
		kb.nwords = nwords; kb.nreads = nreads; kb.ord_base = c->push_ord_base; kb.ord_stride = c->push_ord_stride; kb.maxlen = maxlen;
		kb.d_words = (uint32_t *)keep_alloc(c, nwords * sizeof(uint32_t));
		kb.d_offs = (uint64_t *)keep_alloc(c, (nreads + 1) * sizeof(uint64_t));
		if (!kb.d_words || !kb.d_offs)
			return fail(SDT_ENOMEM, "kept reads: no device memory for another batch (%zu slabs held); run with --host-map", c->keep_slabs.size());
		c->kept.push_back(kb);
		dw = kb.d_words;
		dof = kb.d_offs;
	} else {
		// the kernels that last read this staging buffer must be done before it is overwritten (NSTAGE batches ago): its batch
		// has left the queue (drain_staged never lets the queue grow to NSTAGE) and its buf_free event is the newest one
		if (c->staged.size() - c->staged_head + 1 >= (size_t)sdt_ctx::NSTAGE) {
			const int rcd = drain_staged(c, true);
			if (rcd != SDT_OK) return rcd;
		}
		HIPCHK(hipEventSynchronize(c->buf_free[b]));
		// (capacities with head room: the chunks of a file differ by a few words, and every batch that sets a new record would
		// otherwise cost a hipFree -- a device-wide synchronisation -- and a hipMalloc: 200 M reads in 32-MiB chunks spent
		// seconds there)
		if (c->cap_words[b] < nwords) {
			if (c->d_words[b]) HIPCHK(hipFree(c->d_words[b]));
			c->d_words[b] = nullptr;
			c->cap_words[b] = 0;
			const uint64_t cap = nwords + nwords / 8 + (1u << 16);
			HIPCHK(hipMalloc((void **)&c->d_words[b], cap * sizeof(uint32_t)));
			c->cap_words[b] = cap;
		}
		if (c->cap_offs[b] < nreads + 1) {
			if (c->d_offs[b]) HIPCHK(hipFree(c->d_offs[b]));
			c->d_offs[b] = nullptr;
			c->cap_offs[b] = 0;
			const uint64_t cap = nreads + 1 + nreads / 8 + (1u << 12);
			HIPCHK(hipMalloc((void **)&c->d_offs[b], cap * sizeof(uint64_t)));
			c->cap_offs[b] = cap;
		}
		dw = c->d_words[b];
		dof = c->d_offs[b];
	}
	HIPCHK(hipMemcpyAsync(dw, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
	if (!fixed_len)
		HIPCHK(hipMemcpyAsync(dof, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipEventRecord(c->copied[b], c->copy_stream));
	if (c->staged_head == c->staged.size()) {
		c->staged.clear();
		c->staged_head = 0;
	}
	c->staged.push_back(sdt_ctx::Staged{dw, dof, nreads, maxlen, c->push_ord_base, c->push_ord_stride, b, fixed_len});
	c->push_ord_base += nreads * c->push_ord_stride;     // the next batch continues the read stream
	c->kmers_total_host += kmers;
	c->next_buf = (b + 1) % sdt_ctx::NSTAGE;
	c->push_ticket++;
	if (ticket) *ticket = c->push_ticket;
	return drain_staged(c, false);
}

// would launching this batch make the locality pipeline flush (= block the host)?  (an estimate: sk_scatter decides)
static bool launch_would_flush(const sdt_ctx *c, const sdt_ctx::Staged &b)
{
	const sdt_ctx::SkState &k = c->sk;
	if (!k.ready || (c->flags & SDT_FLAG_DIRECT) || b.maxlen < (uint64_t)c->K + 1)
		return false;
	const uint64_t per_read = b.maxlen - c->K + 1;
	return k.pending_kmers + (b.nreads + SK_TILE_READS) * per_read > sk_batch_limit(c);
}

// launch the kernels of queued batches, oldest first; a launch that would flush waits for STAGE_AHEAD queued copies unless
// `force` (a sync point) or the ring is about to run out of slots
static int drain_staged(sdt_ctx *c, bool force)
{
	if (c->draining)
		return SDT_OK;
	c->draining = true;
	int rc = SDT_OK;
	bool launched = false;
	while (rc == SDT_OK && c->staged_head < c->staged.size()) {
		const sdt_ctx::Staged b = c->staged[c->staged_head];
		const size_t queued = c->staged.size() - c->staged_head;
		if (!force && queued < (size_t)sdt_ctx::STAGE_AHEAD && queued + 2 < (size_t)sdt_ctx::NSTAGE && launch_would_flush(c, b))
			break;
		c->staged_head++;
		launched = true;
		hipError_t e = hipStreamWaitEvent(c->stream, c->copied[b.slot], 0);
		if (e != hipSuccess) { rc = fail(SDT_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; }
		c->ord_base = b.ord_base;
		c->ord_stride = b.ord_stride;
		if (b.fixed_len) {
			hipLaunchKernelGGL(k_fixed_offsets, dim3(scan_grid(c, b.nreads + 1)), dim3(256), 0, c->stream, b.dof, b.nreads, b.fixed_len);
			e = hipGetLastError();
			if (e != hipSuccess) { rc = fail(SDT_EHIP, "k_fixed_offsets: %s", hipGetErrorString(e)); break; }
		}
		rc = launch_count(c, b.dw, b.dof, b.nreads, b.maxlen);
		if (rc != SDT_OK) break;
		e = hipEventRecord(c->buf_free[b.slot], c->stream);
		if (e != hipSuccess) { rc = fail(SDT_EHIP, "hipEventRecord: %s", hipGetErrorString(e)); break; }
	}
	if (launched && c->staged_head == c->staged.size()) {     // the launch cursor has caught up with the push cursor
		c->ord_base = c->push_ord_base;
		c->ord_stride = c->push_ord_stride;
	}
	if (!launched && c->staged_head == c->staged.size()) {    // nothing was queued: calls that count device-resident reads move the
		c->push_ord_base = c->ord_base;                       // launch cursor on their own, and the push cursor follows it
		c->push_ord_stride = c->ord_stride;
	}
	c->draining = false;
	return rc;
}

int sdt_gpu_push_reads_async(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                             uint64_t *ticket)
{
	return push_reads_enqueue(c, packed_words, nwords, offsets, nreads, ticket);
}

int sdt_gpu_push_reads_fixed_async(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, uint64_t nreads, uint64_t read_len,
                                   uint64_t *ticket)
{
	if (read_len == 0)
		return fail(SDT_EINVAL, "read_len must be > 0");
	return push_reads_enqueue(c, packed_words, nwords, nullptr, nreads, ticket, read_len);
}

int sdt_gpu_push_wait(sdt_ctx *c, uint64_t ticket)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (ticket == 0 || ticket > c->push_ticket)
		return ticket == 0 ? SDT_OK : fail(SDT_EINVAL, "ticket %llu was never issued", (unsigned long long)ticket);
	HIPCHK(hipSetDevice(c->device));
	// (copies run in order on one stream: should the ring slot have been reused since, its event stands for a LATER copy)
	HIPCHK(hipEventSynchronize(c->copied[(ticket - 1) % sdt_ctx::NSTAGE]));
	return SDT_OK;
}

int sdt_gpu_hint_total_kmers(sdt_ctx *c, uint64_t kmers)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	c->expect_kmers = kmers;
	// The node table for the job NOW -- while nothing is in it (growing it moves nothing) and BEFORE the pools of the locality pipeline
	// take their share of what is free: pools sized for the announced job on a device whose table then doubles several times left the
	// table's rebuilds fighting the pools for memory (200 M reads: count stage 0.3 -> 1-3 s, and every later table scan crawled).
	// One distinct node per 28 k-mers is what deep transcriptome data gives (C3: 35); data that repeats less grows the table as before.
	if (kmers && c->distinct_known == 0 && c->kmers_since_sync == 0 && c->kmers_total_host == 0 && !c->sk.ready && c->staged_head == c->staged.size()) {
		HIPCHK(hipSetDevice(c->device));
		size_t free_b = 0, total_b = 0;
		HIPCHK(sdti::mem_info(&free_b, &total_b));
		uint64_t est = kmers / 28;
		const uint64_t per_slot = entry_bytes(c->nw) + 4 + ((c->flags & SDT_FLAG_TRACK_FIRST) ? 8 : 0);
		while (est > (1u << 20) && (double)flat_slots_for(est) * (double)per_slot > (double)free_b * 0.3)
			est /= 2;                                    // (never more than ~30 % of what is free)
		if ((double)est > (double)c->slots * MAX_LOAD) {
			const int rc = grow_table(c, est);
			if (rc != SDT_OK) return rc;
		}
	}
	return SDT_OK;
}

int sdt_gpu_push_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets,
                       uint64_t nreads)
{
	uint64_t t = 0;
	const int rc = push_reads_enqueue(c, packed_words, nwords, offsets, nreads, &t);
	if (rc != SDT_OK || nreads == 0)
		return rc;
	return sdt_gpu_push_wait(c, t);                  // the caller may reuse its buffers once the H2D copies have left them
}

int sdt_gpu_count_reads_device(sdt_ctx *c, const void *d_packed_words, uint64_t nwords, const void *d_offsets,
                               uint64_t nreads, uint64_t max_read_len)
{
	(void)nwords;
	if (!c || !d_packed_words || !d_offsets)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_read_len == 0)
		return fail(SDT_EINVAL, "max_read_len must be > 0");
	HIPCHK(hipSetDevice(c->device));
	const int rcd = drain_staged(c, true);               // (batches pushed earlier come first)
	if (rcd != SDT_OK) return rcd;
	return launch_count(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, nreads, max_read_len);
}

int sdt_gpu_finish_count(sdt_ctx *c, uint64_t *kmers_processed, uint64_t *nodes)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (kmers_processed) *kmers_processed = c->h_stats->kmers;
	if (nodes) *nodes = c->h_stats->distinct;
	return SDT_OK;
}

int sdt_gpu_delow(sdt_ctx *c, int d, uint64_t *removed)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (d < 0)
		d = 0;       // pregraph.c:159: negative -d becomes 0
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_delow<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), (uint32_t)d, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_delow<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), (uint32_t)d, c->d_stats);
	else hipLaunchKernelGGL(k_delow<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), (uint32_t)d, c->d_stats);
	HIPCHK(hipGetLastError());
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (removed) *removed = c->h_stats->scratch;
	return SDT_OK;
}

int sdt_gpu_mark_and_hist(sdt_ctx *c, int64_t hist[257], uint64_t *linear)
{
	if (!c || !hist)
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	HIPCHK(hipMemsetAsync(c->d_hist, 0, 257 * sizeof(unsigned long long), c->stream));
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_mark_hist<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), c->d_hist, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_mark_hist<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), c->d_hist, c->d_stats);
	else hipLaunchKernelGGL(k_mark_hist<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), c->d_hist, c->d_stats);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(hist, c->d_hist, 257 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (linear) *linear = c->h_stats->scratch;
	return SDT_OK;
}

int sdt_gpu_export_nodes(sdt_ctx *c, uint64_t *keys, uint32_t *l_links, uint32_t *r_flags, uint32_t *count,
                         uint64_t *first, uint64_t max_nodes, uint64_t *n)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	const uint64_t nodes = c->h_stats->distinct;
	if (n) *n = nodes;
	if (!keys && !l_links && !r_flags && !count && !first)
		return SDT_OK;
	if (first && !c->d_first)
		return fail(SDT_ESTATE, "first-occurrence ordinals were not tracked: init with SDT_FLAG_TRACK_FIRST");
	if (max_nodes < nodes)
		return fail(SDT_EINVAL, "export arrays hold %llu nodes, table has %llu", (unsigned long long)max_nodes,
		            (unsigned long long)nodes);
	uint64_t *d_keys = nullptr;
	uint32_t *d_l = nullptr, *d_r = nullptr, *d_c = nullptr;
	uint64_t *d_f = nullptr;
	const uint64_t m = nodes ? nodes : 1;
	int ret = SDT_OK;
#define EXP_CHK(expr)                                                                                  \
	do {                                                                                               \
		hipError_t e3_ = (expr);                                                                       \
		if (e3_ != hipSuccess) {                                                                       \
			ret = fail(e3_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e3_)); \
			goto done;                                                                                 \
		}                                                                                              \
	} while (0)
	if (keys) EXP_CHK(hipMalloc((void **)&d_keys, m * c->nw * sizeof(uint64_t)));
	if (l_links) EXP_CHK(hipMalloc((void **)&d_l, m * sizeof(uint32_t)));
	if (r_flags) EXP_CHK(hipMalloc((void **)&d_r, m * sizeof(uint32_t)));
	if (count) EXP_CHK(hipMalloc((void **)&d_c, m * sizeof(uint32_t)));
	if (first) EXP_CHK(hipMalloc((void **)&d_f, m * sizeof(uint64_t)));
	EXP_CHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	{
		const int g = scan_grid(c, view_slots(c));
		if (c->nw == 1) hipLaunchKernelGGL(k_export<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_export<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
		else hipLaunchKernelGGL(k_export<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
	}
	EXP_CHK(hipGetLastError());
	if (keys) EXP_CHK(hipMemcpyAsync(keys, d_keys, nodes * c->nw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	if (l_links) EXP_CHK(hipMemcpyAsync(l_links, d_l, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (r_flags) EXP_CHK(hipMemcpyAsync(r_flags, d_r, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (count) EXP_CHK(hipMemcpyAsync(count, d_c, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (first) EXP_CHK(hipMemcpyAsync(first, d_f, nodes * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	EXP_CHK(hipStreamSynchronize(c->stream));
done:
	if (d_keys) (void)hipFree(d_keys);
	if (d_l) (void)hipFree(d_l);
	if (d_r) (void)hipFree(d_r);
	if (d_c) (void)hipFree(d_c);
	if (d_f) (void)hipFree(d_f);
	return ret;
}

// ---- second pass: prlRead2edge on the device ---------------------------------------------------------------
int sdt_gpu_load_paths(sdt_ctx *c, const uint64_t *keys, const uint64_t *path_words, uint64_t n, const uint64_t *patch_keys,
                       const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed)
{
	if (!c || (npatch && (!patch_keys || !patch_info)))
		return fail(SDT_EINVAL, "NULL argument");
	const bool by_index = keys == nullptr;
	// keys == NULL and path_words == NULL: the path words sdt_gpu_build_edges left on the device
	uint64_t *d_own = (!keys && !path_words && n) ? sdti::graph_take_path_words(c->gx, n) : nullptr;
	if (n && !path_words && !d_own)
		return fail(keys ? SDT_EINVAL : SDT_ESTATE, "no path words: pass them, or build the edges with sdt_gpu_build_edges first");
	if (by_index && n && (!c->d_idx || c->idx_slots != view_slots(c) || c->idx_n != n)) {
		if (d_own) (void)hipFree(d_own);
		return fail(SDT_ESTATE, "keys == NULL needs the node index of sdt_gpu_set_node_index for the same %llu nodes", (unsigned long long)n);
	}
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	uint64_t *d_k = nullptr, *d_i = nullptr;
	if (n) {
		if (!by_index) HIPCHK(hipMalloc((void **)&d_k, n * c->nw * sizeof(uint64_t)));
		hipError_t e = d_own ? hipSuccess : hipMalloc((void **)&d_i, n * sizeof(uint64_t));
		if (e != hipSuccess) { if (d_k) (void)hipFree(d_k); return fail(SDT_ENOMEM, "path words: %s", hipGetErrorString(e)); }
		if (d_own) d_i = d_own;
		int rcu = by_index ? SDT_OK : sdti::h2d_big(c->copy_stream, d_k, keys, n * c->nw * sizeof(uint64_t));
		if (rcu == SDT_OK && !d_own) rcu = sdti::h2d_big(c->copy_stream, d_i, path_words, n * sizeof(uint64_t));
		if (rcu != SDT_OK) { if (d_k) (void)hipFree(d_k); (void)hipFree(d_i); return rcu; }
		if (by_index) {
			const int g = scan_grid(c, view_slots(c));
			if (c->nw == 1) hipLaunchKernelGGL(k_set_paths_by_index<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), c->d_idx, d_i, n, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_set_paths_by_index<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), c->d_idx, d_i, n, c->d_stats);
			else hipLaunchKernelGGL(k_set_paths_by_index<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), c->d_idx, d_i, n, c->d_stats);
		} else {
			const int g = scan_grid(c, n);
			if (c->nw == 1) hipLaunchKernelGGL(k_set_paths<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_k, d_i, n, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_set_paths<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_k, d_i, n, c->d_stats);
			else hipLaunchKernelGGL(k_set_paths<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_k, d_i, n, c->d_stats);
		}
		hipError_t le = hipGetLastError();
		hipError_t se = hipStreamSynchronize(c->stream);
		if (d_k) (void)hipFree(d_k);
		(void)hipFree(d_i);
		if (le != hipSuccess || se != hipSuccess)
			return fail(SDT_EHIP, "k_set_paths: %s", hipGetErrorString(le != hipSuccess ? le : se));
	}
	int rc = c->nw == 1 ? build_patch_table<1>(c, patch_keys, patch_info, npatch)
	       : c->nw == 2 ? build_patch_table<2>(c, patch_keys, patch_info, npatch)
	                    : build_patch_table<4>(c, patch_keys, patch_info, npatch);
	if (rc != SDT_OK)
		return rc;
	// arcs: a few per edge in practice; the map doubles (and the pass is redone) if it ever fills up
	uint64_t slots = 1 << 16;
	while (slots < 8 * (num_ed + 1))
		slots <<= 1;
	if (c->d_arcs) HIPCHK(hipFree(c->d_arcs));
	c->d_arcs = nullptr;
	HIPCHK(hipMalloc((void **)&c->d_arcs, slots * sizeof(ArcEnt)));
	c->arc_slots = slots;
	rc = sync_stats(c);
	if (rc != SDT_OK)
		return fail(SDT_ESTATE, "sdt_gpu_load_paths: %llu nodes are not in the table", (unsigned long long)c->h_stats->probe_fail);
	c->paths_loaded = true;
	return SDT_OK;
}

int sdt_gpu_export_paths(sdt_ctx *c, uint64_t *keys, uint64_t *path_words, uint64_t max_nodes, uint64_t *n)
{
	if (!c || !n)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->paths_loaded)
		return fail(SDT_ESTATE, "call sdt_gpu_load_paths first");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	const uint64_t nodes = c->h_stats->distinct;
	*n = nodes;
	if (!keys && !path_words)
		return SDT_OK;
	if (!keys || !path_words || max_nodes < nodes)
		return fail(SDT_EINVAL, "export arrays hold %llu nodes, the table has %llu", (unsigned long long)max_nodes, (unsigned long long)nodes);
	uint64_t *d_k = nullptr, *d_p = nullptr;
	const uint64_t m = nodes ? nodes : 1;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	hipError_t e = hipMalloc((void **)&d_p, m * 8);
	if (e != hipSuccess) { (void)hipFree(d_k); return fail(SDT_ENOMEM, "path export: %s", hipGetErrorString(e)); }
	int ret = SDT_OK;
	e = hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream);
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_export_paths<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_export_paths<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	else hipLaunchKernelGGL(k_export_paths<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	if (e == hipSuccess) e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) ret = fail(SDT_EHIP, "k_export_paths: %s", hipGetErrorString(e));
	if (ret == SDT_OK) {
		// every non-empty slot took a place: a table whose counters disagree with its slots would publish an incomplete graph
		unsigned long long placed = 0;
		e = hipMemcpy(&placed, &c->d_stats->scratch, sizeof placed, hipMemcpyDeviceToHost);
		if (e != hipSuccess) ret = fail(SDT_EHIP, "k_export_paths: %s", hipGetErrorString(e));
		else if (placed != nodes)
			ret = fail(SDT_ESTATE, "path export: the table holds %llu nodes, its counters say %llu", placed, (unsigned long long)nodes);
	}
	if (ret == SDT_OK) ret = sdti::d2h_big(c->copy_stream, keys, d_k, nodes * c->nw * 8);
	if (ret == SDT_OK) ret = sdti::d2h_big(c->copy_stream, path_words, d_p, nodes * 8);
	(void)hipFree(d_k);
	(void)hipFree(d_p);
	return ret;
}

int sdt_gpu_import_paths(sdt_ctx *c, const uint64_t *keys, const uint64_t *path_words, uint64_t n, const uint64_t *patch_keys,
                         const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed)
{
	if (!c || (n && (!keys || !path_words)) || (npatch && (!patch_keys || !patch_info)))
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	// the table of this rank's shard makes way (the reads kept for the second pass stay): an empty flat table with room for the graph
	if (c->d_idx) { (void)hipFree(c->d_idx); c->d_idx = nullptr; c->idx_slots = c->idx_n = 0; }
	const uint64_t want = flat_slots_for(n);
	if (want > c->slots) {
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->d_ent) (void)hipFree(c->d_ent);
		if (c->d_aux) (void)hipFree(c->d_aux);
		if (c->d_first) (void)hipFree(c->d_first);
		c->d_ent = nullptr; c->d_aux = nullptr; c->d_first = nullptr;
		rc = alloc_table(c, want, &c->d_ent, &c->d_aux, &c->d_first);
		if (rc != SDT_OK) { c->slots = 0; return rc; }
		c->slots = want;
	}
	rc = launch_clear(c, c->d_ent, c->d_aux, c->d_first, c->slots);
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemsetAsync(&c->d_stats->distinct, 0, sizeof(unsigned long long), c->stream));
	c->distinct_known = 0;
	c->kmers_since_sync = c->hard_since_sync = 0;
	uint64_t *d_k = nullptr, *d_p = nullptr;
	const uint64_t STEP = 1ULL << 26;                // nodes per upload: bounded staging memory
	const uint64_t m = n < STEP ? (n ? n : 1) : STEP;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	hipError_t e = hipMalloc((void **)&d_p, m * 8);
	if (e != hipSuccess) { (void)hipFree(d_k); return fail(SDT_ENOMEM, "path import: %s", hipGetErrorString(e)); }
	for (uint64_t i0 = 0; i0 < n && rc == SDT_OK; i0 += STEP) {
		const uint64_t k = n - i0 < STEP ? n - i0 : STEP;
		rc = sdti::h2d_big(c->copy_stream, d_k, keys + i0 * c->nw, k * c->nw * 8);
		if (rc == SDT_OK) rc = sdti::h2d_big(c->copy_stream, d_p, path_words + i0, k * 8);
		if (rc != SDT_OK) break;
		const int g = scan_grid(c, k);
		if (c->nw == 1) hipLaunchKernelGGL(k_import_paths<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d_k, d_p, k, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_import_paths<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d_k, d_p, k, c->d_stats);
		else hipLaunchKernelGGL(k_import_paths<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d_k, d_p, k, c->d_stats);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) rc = fail(SDT_EHIP, "k_import_paths: %s", hipGetErrorString(e));
	}
	(void)hipFree(d_k);
	(void)hipFree(d_p);
	if (rc != SDT_OK) return rc;
	// patch table, arc map, and the check that every key went in once (sdt_gpu_load_paths with no node of its own to set)
	rc = sdt_gpu_load_paths(c, nullptr, nullptr, 0, patch_keys, patch_info, npatch, num_ed);
	if (rc == SDT_OK && c->h_stats->distinct != n)
		return fail(SDT_ESTATE, "path import: %llu keys came, the table holds %llu nodes (a key twice?)", (unsigned long long)n,
		            (unsigned long long)c->h_stats->distinct);
	return rc;
}

int sdt_gpu_release_table(sdt_ctx *c)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sdti::release_pass1(c);                 // drains pass 1; the pools go back
	if (rc != SDT_OK) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->d_ent) (void)hipFree(c->d_ent);
	if (c->d_aux) (void)hipFree(c->d_aux);
	if (c->d_first) (void)hipFree(c->d_first);
	if (c->d_idx) { (void)hipFree(c->d_idx); c->d_idx = nullptr; c->idx_slots = c->idx_n = 0; }
	c->d_ent = nullptr; c->d_aux = nullptr; c->d_first = nullptr;
	c->slots = 0;
	HIPCHK(hipMemsetAsync(&c->d_stats->distinct, 0, sizeof(unsigned long long), c->stream));
	c->distinct_known = 0;
	c->kmers_since_sync = c->hard_since_sync = 0;
	return SDT_OK;
}

int sdt_gpu_map_reads(sdt_ctx *c, uint64_t *reads_processed, uint64_t *arcs)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (!c->paths_loaded)
		return fail(SDT_ESTATE, "call sdt_gpu_load_paths first");
	if (!(c->flags & SDT_FLAG_KEEP_READS) && c->kept.empty())
		return fail(SDT_ESTATE, "the reads were not kept: init with SDT_FLAG_KEEP_READS (or hand them over with sdt_gpu_keep_reads)");
	HIPCHK(hipSetDevice(c->device));
	for (int attempt = 0; attempt < 8; attempt++) {
		// ArcEnt.first starts at ~0 (atomicMin), key/mult at 0
		HIPCHK(hipMemsetAsync(c->d_arcs, 0, c->arc_slots * sizeof(ArcEnt), c->stream));
		HIPCHK(hipMemset2DAsync(&c->d_arcs[0].first, sizeof(ArcEnt), 0xFF, sizeof(unsigned long long), c->arc_slots, c->stream));
		HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
		uint64_t reads = 0;
		// One lane per read and ~120 dependent look-ups per lane: a kept batch of the CLI (10^5 reads) is 1 600 waves, six per CU, and
		// its launch lasts as long as the longest chain (0.4 ms: 1 900 launches one after the other took 770 ms at 200 M reads).  The
		// batches are independent (arcs are atomic adds / mins): several streams keep several launches on the device at a time.
		constexpr int NS = 6;
		hipStream_t ms[NS];
		hipEvent_t ready, done[NS];
		int ns = c->kept.size() > 8 ? NS : 1;
		if (ns > 1) {
			if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) ns = 1;
			for (int i = 0; i < ns && ns > 1; i++)
				if (hipStreamCreateWithFlags(&ms[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
					return fail(SDT_EHIP, "streams for the second read pass");
		}
		if (ns > 1) {
			HIPCHK(hipEventRecord(ready, c->stream));
			for (int i = 0; i < ns; i++) HIPCHK(hipStreamWaitEvent(ms[i], ready, 0));
		}
		size_t bi = 0;
		for (auto &kb : c->kept) {
			const int g = scan_grid(c, kb.nreads);
			const hipStream_t st = ns > 1 ? ms[bi++ % (size_t)ns] : c->stream;
			if (c->nw == 1) hipLaunchKernelGGL(k_map_reads<1>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<1>(c), (const PatchEnt<1> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_map_reads<2>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<2>(c), (const PatchEnt<2> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			else hipLaunchKernelGGL(k_map_reads<4>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<4>(c), (const PatchEnt<4> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			HIPCHK(hipGetLastError());
			reads += kb.nreads;
		}
		if (ns > 1) {
			for (int i = 0; i < ns; i++) { HIPCHK(hipEventRecord(done[i], ms[i])); HIPCHK(hipStreamWaitEvent(c->stream, done[i], 0)); }
		}
		HIPCHK(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (ns > 1) {
			for (int i = 0; i < ns; i++) { (void)hipStreamDestroy(ms[i]); (void)hipEventDestroy(done[i]); }
			(void)hipEventDestroy(ready);
		}
		if (c->h_stats->scratch)
			return fail(SDT_ESTATE, "%llu reads hold a k-mer that is not in the node table (different reads than pass 1?)",
			            (unsigned long long)c->h_stats->scratch);
		if (c->h_stats->probe_fail == 0) {
			if (reads_processed) *reads_processed = reads;
			if (arcs) {
				// count occupied slots by exporting nothing but the cursor
				unsigned long long *d_cur = nullptr;
				HIPCHK(hipMalloc((void **)&d_cur, sizeof(unsigned long long)));
				HIPCHK(hipMemsetAsync(d_cur, 0, sizeof(unsigned long long), c->stream));
				hipLaunchKernelGGL(k_export_arcs, dim3(scan_grid(c, c->arc_slots)), dim3(TPB), 0, c->stream, c->d_arcs, c->arc_slots, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, 0ULL, d_cur);
				unsigned long long h = 0;
				HIPCHK(hipMemcpyAsync(&h, d_cur, sizeof h, hipMemcpyDeviceToHost, c->stream));
				HIPCHK(hipStreamSynchronize(c->stream));
				(void)hipFree(d_cur);
				*arcs = h;
			}
			return SDT_OK;
		}
		// arc map too small: double it and redo the pass (arc adds are idempotent only from a clean map)
		HIPCHK(hipMemsetAsync(&c->d_stats->probe_fail, 0, sizeof(unsigned long long), c->stream));
		HIPCHK(hipFree(c->d_arcs));
		c->d_arcs = nullptr;
		c->arc_slots <<= 1;
		HIPCHK(hipMalloc((void **)&c->d_arcs, c->arc_slots * sizeof(ArcEnt)));
	}
	return fail(SDT_EFULL, "arc map keeps overflowing");
}

int sdt_gpu_export_arcs(sdt_ctx *c, uint32_t *from, uint32_t *to, uint32_t *mult, uint64_t *first, uint64_t max_arcs, uint64_t *n)
{
	if (!c || !from || !to || !mult || !first)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->d_arcs)
		return fail(SDT_ESTATE, "no arcs: call sdt_gpu_map_reads first");
	HIPCHK(hipSetDevice(c->device));
	uint32_t *d_f = nullptr, *d_t = nullptr, *d_m = nullptr;
	uint64_t *d_o = nullptr;
	unsigned long long *d_cur = nullptr;
	const uint64_t m = max_arcs ? max_arcs : 1;
	int ret = SDT_OK;
	unsigned long long h = 0;
#define ARC_CHK(expr) do { hipError_t e4_ = (expr); if (e4_ != hipSuccess) { ret = fail(SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e4_)); goto done; } } while (0)
	ARC_CHK(hipMalloc((void **)&d_f, m * 4));
	ARC_CHK(hipMalloc((void **)&d_t, m * 4));
	ARC_CHK(hipMalloc((void **)&d_m, m * 4));
	ARC_CHK(hipMalloc((void **)&d_o, m * 8));
	ARC_CHK(hipMalloc((void **)&d_cur, 8));
	ARC_CHK(hipMemsetAsync(d_cur, 0, 8, c->stream));
	hipLaunchKernelGGL(k_export_arcs, dim3(scan_grid(c, c->arc_slots)), dim3(TPB), 0, c->stream, c->d_arcs, c->arc_slots, d_f, d_t, d_m, d_o, (unsigned long long)max_arcs, d_cur);
	ARC_CHK(hipGetLastError());
	ARC_CHK(hipMemcpyAsync(&h, d_cur, 8, hipMemcpyDeviceToHost, c->stream));
	ARC_CHK(hipStreamSynchronize(c->stream));
	if (h > max_arcs) { ret = fail(SDT_EINVAL, "arc arrays hold %llu, need %llu", (unsigned long long)max_arcs, h); goto done; }
	// (in the order *.preArc lists them: the host's own sort finds nothing left to do)
	ret = sdti::sort_arcs_for_output(c->stream, c->cu_count, d_f, d_t, d_m, d_o, h);
	if (ret != SDT_OK) goto done;
	ARC_CHK(hipMemcpy(from, d_f, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(to, d_t, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(mult, d_m, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(first, d_o, h * 8, hipMemcpyDeviceToHost));
	if (n) *n = h;
done:
	if (d_f) (void)hipFree(d_f);
	if (d_t) (void)hipFree(d_t);
	if (d_m) (void)hipFree(d_m);
	if (d_o) (void)hipFree(d_o);
	if (d_cur) (void)hipFree(d_cur);
	return ret;
}

// Gigabytes from PAGEABLE host memory (the host graph's arrays: 5.4 GB of keys, as much again of path words at 200 M reads): the
// runtime stages such a copy through one small pinned buffer on one thread, ~2.2 GB/s (2.4 s per array).  Here: two pinned
// buffers of 64 MiB, filled by four threads while the other one is on the wire.  Synchronous: the source may be freed on return.
// ---- map stage: prlContig2nodes / prlRead2Ctg ----------------------------------------------------------------
static int ab_reserve(sdt_ctx *c, int i, size_t bytes)
{
	if (c->ab_cap[i] >= bytes) return SDT_OK;
	if (c->ab[i]) HIPCHK(hipFree(c->ab[i]));
	c->ab[i] = nullptr;
	c->ab_cap[i] = 0;
	const size_t want = bytes + bytes / 4 + 256;
	hipError_t e = hipMalloc(&c->ab[i], want);
	if (e != hipSuccess) return fail(SDT_ENOMEM, "map staging (%zu bytes): %s", want, hipGetErrorString(e));
	c->ab_cap[i] = want;
	return SDT_OK;
}

int sdt_gpu_index_contigs(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, const uint32_t *ids,
                          uint64_t ncontigs)
{
	if (!c || !packed_words || !offsets || !ids)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX to index contigs");
	if (c->index_final)
		return fail(SDT_ESTATE, "the contig index is final once reads have been aligned: sdt_gpu_reset to start over");
	if (ncontigs == 0)
		return SDT_OK;
	uint64_t kmers = 0;
	for (uint64_t i = 0; i < ncontigs; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at contig %llu", (unsigned long long)i);
		const uint64_t len = offsets[i + 1] - offsets[i];
		if (len >= (1ULL << CTG_POS_BITS))
			return fail(SDT_EINVAL, "contig %llu is %llu bases long: positions are 24-bit (kmer_t.r_links)", (unsigned long long)i, (unsigned long long)len);
		if (len >= (uint64_t)c->K) kmers += len - c->K + 1;
	}
	if (((offsets[ncontigs] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words", (unsigned long long)(((offsets[ncontigs] + 15) >> 4) + TAIL_PAD), TAIL_PAD);
	HIPCHK(hipSetDevice(c->device));
	// contig ordinal -> id table grows by this batch
	if (c->ctg_ord + ncontigs > c->ctg_ids_cap) {
		const uint64_t cap = (c->ctg_ord + ncontigs) * 2 + 1024;
		uint32_t *n = nullptr;
		HIPCHK(hipMalloc((void **)&n, cap * sizeof(uint32_t)));
		if (c->ctg_ord) HIPCHK(hipMemcpyAsync(n, c->d_ctg_ids, c->ctg_ord * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->d_ctg_ids) HIPCHK(hipFree(c->d_ctg_ids));
		c->d_ctg_ids = n;
		c->ctg_ids_cap = cap;
	}
	int rc = ab_reserve(c, 0, nwords * sizeof(uint32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 1, (ncontigs + 1) * sizeof(uint64_t));
	if (rc != SDT_OK) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[0], packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[1], offsets, (ncontigs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->d_ctg_ids + c->ctg_ord, ids, ncontigs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	rc = ensure_room(c, kmers);
	if (rc != SDT_OK) return rc;
	const int g = scan_grid(c, offsets[ncontigs] ? offsets[ncontigs] : 1);
	const uint32_t *dw = (const uint32_t *)c->ab[0];
	const uint64_t *dof = (const uint64_t *)c->ab[1];
	if (c->nw == 1) hipLaunchKernelGGL(k_index_contigs<1>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<1>(c), c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_index_contigs<2>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<2>(c), c->d_stats);
	else hipLaunchKernelGGL(k_index_contigs<4>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<4>(c), c->d_stats);
	HIPCHK(hipGetLastError());
	c->kmers_since_sync += kmers;
	c->ctg_ord += ncontigs;
	HIPCHK(hipStreamSynchronize(c->stream));     // the caller may reuse its buffers
	return SDT_OK;
}

int sdt_gpu_set_contig_table(sdt_ctx *c, const uint32_t *length, const uint32_t *twin, uint64_t num_ctg)
{
	if (!c || !length || !twin)
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->d_ctg_len) HIPCHK(hipFree(c->d_ctg_len));
	if (c->d_ctg_twin) HIPCHK(hipFree(c->d_ctg_twin));
	c->d_ctg_len = c->d_ctg_twin = nullptr;
	HIPCHK(hipMalloc((void **)&c->d_ctg_len, (num_ctg + 1) * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&c->d_ctg_twin, (num_ctg + 1) * sizeof(uint32_t)));
	HIPCHK(hipMemcpy(c->d_ctg_len, length, (num_ctg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(c->d_ctg_twin, twin, (num_ctg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
	c->num_ctg = num_ctg;
	return SDT_OK;
}

static int launch_align(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads, uint64_t max_read_len,
                        const int32_t *d_align_len, int align_len_all, uint64_t *d_info, Hit *d_hits, uint64_t max_hits)
{
	if (!c->d_ctg_len)
		return fail(SDT_ESTATE, "call sdt_gpu_set_contig_table first");
	if (max_read_len < (uint64_t)c->K + 1) max_read_len = (uint64_t)c->K + 1;
	const int max_kmers = (int)(max_read_len - c->K + 1);
	const size_t per_wave = ((size_t)max_kmers + 2 * MAX_HITS) * sizeof(uint64_t);
	int waves = 4;
	while (waves > 1 && per_wave * waves > 48 * 1024) waves >>= 1;
	if (per_wave > 64 * 1024)
		return fail(SDT_EINVAL, "reads of %llu bases do not fit the per-wavefront LDS window", (unsigned long long)max_read_len);
	if (!c->index_final) {
		int rcs = sync_stats(c);                          // counts stay readable through finish_count (host copy)
		if (rcs != SDT_OK) return rcs;
		const int g = scan_grid(c, view_slots(c));
		if (c->nw == 1) hipLaunchKernelGGL(k_finalize_contig_index<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_finalize_contig_index<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		else hipLaunchKernelGGL(k_finalize_contig_index<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		HIPCHK(hipGetLastError());
		c->index_final = true;
	}
	if (!c->d_hit_cursor) HIPCHK(hipMalloc((void **)&c->d_hit_cursor, sizeof(unsigned long long)));
	{
		const unsigned long long first_extra = nreads;      // hits[0 .. nreads) = first hit of each read, the rest follows
		HIPCHK(hipMemcpyAsync(c->d_hit_cursor, &first_extra, sizeof first_extra, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	uint64_t blocks = (nreads + waves - 1) / waves;
	const uint64_t cap = (uint64_t)c->cu_count * 32;
	if (blocks > cap) blocks = cap;
	if (blocks == 0) blocks = 1;
	EventPair *ev = next_event(c);
	if (ev) HIPCHK(hipEventRecord(ev->a, c->stream));
#define ALIGN_LAUNCH(NWV) hipLaunchKernelGGL(k_align_reads<NWV>, dim3((unsigned)blocks), dim3(TPB), per_wave * waves, c->stream, d_words, d_offs, nreads, \
	d_align_len, align_len_all, c->K, table_of<NWV>(c), (const uint32_t *)c->d_ctg_len, \
	(const uint32_t *)c->d_ctg_twin, c->num_ctg, max_kmers, waves, d_info, d_hits, (unsigned long long)max_hits, c->d_hit_cursor, c->d_stats)
	if (c->nw == 1) ALIGN_LAUNCH(1);
	else if (c->nw == 2) ALIGN_LAUNCH(2);
	else ALIGN_LAUNCH(4);
#undef ALIGN_LAUNCH
	HIPCHK(hipGetLastError());
	if (ev) {
		HIPCHK(hipEventRecord(ev->b, c->stream));
		ev->kmers = 0;
	}
	return SDT_OK;
}

int sdt_gpu_align_reads_device(sdt_ctx *c, const void *d_packed_words, const void *d_offsets, uint64_t nreads, uint64_t max_read_len,
                               const void *d_align_len, int align_len_all, void *d_read_info, void *d_hits, uint64_t max_hits,
                               uint64_t *nhits)
{
	if (!c || !d_packed_words || !d_offsets || !d_read_info || !d_hits)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_hits < nreads)
		return fail(SDT_EINVAL, "hits[] must hold at least one entry per read (%llu < %llu)", (unsigned long long)max_hits, (unsigned long long)nreads);
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX");
	HIPCHK(hipSetDevice(c->device));
	int rc = launch_align(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, nreads, max_read_len,
	                      (const int32_t *)d_align_len, align_len_all, (uint64_t *)d_read_info, (Hit *)d_hits, max_hits);
	if (rc != SDT_OK) return rc;
	unsigned long long h = 0;
	HIPCHK(hipMemcpyAsync(&h, c->d_hit_cursor, sizeof h, hipMemcpyDeviceToHost, c->stream));
	rc = sync_stats(c);
	if (rc != SDT_OK)
		return fail(SDT_ESTATE, "sdt_gpu_align_reads: %llu reads are longer than max_read_len or hit a contig outside the contig table",
		            (unsigned long long)c->h_stats->probe_fail);
	if (nhits) *nhits = h;
	if (h > max_hits)
		return fail(SDT_EFULL, "hit array holds %llu, the batch produced %llu", (unsigned long long)max_hits, h);
	return SDT_OK;
}

int sdt_gpu_align_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                        const int32_t *align_len, int align_len_all, uint64_t *read_info, sdt_hit *hits, uint64_t max_hits,
                        uint64_t *nhits)
{
	if (!c || !packed_words || !offsets || !read_info || (!hits && max_hits))
		return fail(SDT_EINVAL, "NULL argument");
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX");
	if (nreads == 0) { if (nhits) *nhits = 0; return SDT_OK; }
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)i);
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	}
	if (((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words", (unsigned long long)(((offsets[nreads] + 15) >> 4) + TAIL_PAD), TAIL_PAD);
	HIPCHK(hipSetDevice(c->device));
	int rc = ab_reserve(c, 0, nwords * sizeof(uint32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 1, (nreads + 1) * sizeof(uint64_t));
	if (rc == SDT_OK && align_len) rc = ab_reserve(c, 2, nreads * sizeof(int32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 3, nreads * sizeof(uint64_t));
	if (max_hits < nreads)
		return fail(SDT_EINVAL, "hits[] must hold at least one entry per read (%llu < %llu)", (unsigned long long)max_hits, (unsigned long long)nreads);
	if (rc == SDT_OK) rc = ab_reserve(c, 4, (max_hits ? max_hits : 1) * sizeof(Hit));
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemcpyAsync(c->ab[0], packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[1], offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	if (align_len) HIPCHK(hipMemcpyAsync(c->ab[2], align_len, nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	uint64_t got = 0;
	rc = sdt_gpu_align_reads_device(c, c->ab[0], c->ab[1], nreads, maxlen, align_len ? c->ab[2] : nullptr, align_len_all, c->ab[3], c->ab[4],
	                                max_hits, &got);
	if (nhits) *nhits = got;
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemcpyAsync(read_info, c->ab[3], nreads * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	if (got) HIPCHK(hipMemcpyAsync(hits, c->ab[4], got * sizeof(Hit), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return SDT_OK;
}

int sdt_gpu_kernel_time(sdt_ctx *c, int reset, double *ms, uint64_t *launches, uint64_t *kmers)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	double total = 0;
	uint64_t km = 0;
	for (size_t i = 0; i < c->ev_used; i++) {
		float t = 0;
		HIPCHK(hipEventElapsedTime(&t, c->ev[i].a, c->ev[i].b));
		total += t;
		km += c->ev[i].kmers;
	}
	if (ms) *ms = total;
	if (launches) *launches = c->ev_used;
	if (kmers) *kmers = km;
	if (reset)
		c->ev_used = 0;
	return SDT_OK;
}

// ---- multi-GPU ----------------------------------------------------------------------------------------------
int sdt_gpu_comm_id(sdt_comm_id *id)
{
	if (!id)
		return fail(SDT_EINVAL, "NULL argument");
	int rc = rccl_load();
	if (rc != SDT_OK)
		return rc;
	static_assert(sizeof(sdt_comm_id) == sizeof(NcclId), "ncclUniqueId is 128 bytes");
	NCCLCHK(g_rccl.GetUniqueId((NcclId *)id));
	return SDT_OK;
}

int sdt_gpu_comm_init(sdt_ctx *c, const sdt_comm_id *id, int rank, int nranks)
{
	if (!c || !id || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	if (c->comm.kind)
		return fail(SDT_ESTATE, "the context already has a communicator");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.open_rccl((const NcclId *)id, rank, nranks);
}

int sdt_gpu_comm_init_shm(sdt_ctx *c, const char *name, int rank, int nranks)
{
	if (!c || !name || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	if (c->comm.kind)
		return fail(SDT_ESTATE, "the context already has a communicator");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.open_shm(name, rank, nranks, true);
}

int sdt_comm_selftest_shm(const char *name, int rank, int nranks, int rounds)
{
	// host-only exercise of the shared-memory transport's control plane (no device): what the CPU tests run with
	// several processes -- all-gather, all-reduce and barriers must agree on every rank, round after round
	Comm cm;
	int rc = cm.open_shm(name, rank, nranks, false);
	if (rc != SDT_OK)
		return rc;
	for (int it = 0; it < rounds && rc == SDT_OK; it++) {
		std::vector<uint32_t> mine(257), all((size_t)257 * nranks);
		for (int i = 0; i < 257; i++) mine[i] = (uint32_t)(rank * 1000003 + it * 7919 + i);
		rc = cm.allgather_host(mine.data(), all.data(), 257 * sizeof(uint32_t));
		for (int r = 0; r < nranks && rc == SDT_OK; r++)
			for (int i = 0; i < 257; i++)
				if (all[(size_t)r * 257 + i] != (uint32_t)(r * 1000003 + it * 7919 + i))
					rc = fail(SDT_EHIP, "all-gather: rank %d got a wrong word from rank %d in round %d", rank, r, it);
		int64_t v[3] = {rank + 1, it, (int64_t)1 << 40};
		if (rc == SDT_OK) rc = cm.allreduce_sum_host(v, 3);
		if (rc == SDT_OK && (v[0] != (int64_t)nranks * (nranks + 1) / 2 || v[1] != (int64_t)it * nranks || v[2] != ((int64_t)nranks << 40)))
			rc = fail(SDT_EHIP, "all-reduce: rank %d got wrong sums in round %d", rank, it);
	}
	cm.close_all();
	return rc;
}

int sdt_gpu_allreduce_i64(sdt_ctx *c, int64_t *vals, int n)
{
	if (!c || !vals || n < 0 || (size_t)n * sizeof(int64_t) > SHM_CTRL_BYTES)
		return fail(SDT_EINVAL, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.allreduce_sum_host(vals, n);
}

int sdt_gpu_comm_stats(sdt_ctx *c, uint64_t *bytes_sent, uint64_t *bytes_recv, double *exchange_ms, uint64_t *exchanges)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	if (c->comm.xstream)
		HIPCHK(hipStreamSynchronize(c->comm.xstream));
	c->comm.harvest_time();
	if (bytes_sent) *bytes_sent = c->comm.bytes_sent;
	if (bytes_recv) *bytes_recv = c->comm.bytes_recv;
	if (exchange_ms) *exchange_ms = c->comm.exchange_ms;
	if (exchanges) *exchanges = c->comm.exchanges;
	return SDT_OK;
}

int sdt_gpu_shard_ranges(const sdt_ctx *c, uint32_t *first_bucket)
{
	if (!c || !first_bucket)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->sh.have_ranges)
		return fail(SDT_ESTATE, "no sharded call yet: the bucket ranges are cut on the first one");
	for (int r = 0; r <= c->comm.nranks; r++)
		first_bucket[r] = c->sh.ranges[r];
	return SDT_OK;
}

int sdt_shard_cut_ranges(const uint32_t *mat, int nranks, uint32_t *ranges)
{
	if (!mat || !ranges || nranks < 1 || nranks > SHARD_MAX_RANKS)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	shard_cut_ranges(mat, nranks, ranges);
	return SDT_OK;
}

int sdt_shard_plan(const uint32_t *mat, int nranks, int me, const uint32_t *ranges, uint32_t recv_chunks, uint32_t t,
                   uint32_t *subrounds, uint32_t *send_begin, uint32_t *send_count, uint32_t *send_at, uint32_t *recv_count,
                   uint32_t *recv_at)
{
	if (!mat || !ranges || !subrounds || !send_begin || !send_count || !send_at || !recv_count || !recv_at)
		return fail(SDT_EINVAL, "NULL argument");
	if (nranks < 1 || nranks > SHARD_MAX_RANKS || me < 0 || me >= nranks || recv_chunks == 0)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks, a receive buffer of at least one chunk)");
	const uint32_t S = shard_subrounds(mat, nranks, ranges, recv_chunks);
	*subrounds = S;
	if (t >= S)
		return fail(SDT_EINVAL, "sub-round %u of %u", t, S);
	ShardRound r;
	shard_round(mat, nranks, me, ranges, t, S, r);
	for (int p = 0; p < nranks; p++) {
		send_begin[p] = r.send_begin[p]; send_count[p] = r.send_count[p]; send_at[p] = r.send_at[p];
		recv_count[p] = r.recv_count[p]; recv_at[p] = r.recv_at[p];
	}
	return SDT_OK;
}

int sdt_sk_plan_count_items(const uint32_t *off2, const uint64_t *kpre2, uint32_t nbuckets, uint64_t first_limit, uint64_t limit,
                            uint32_t max_launches, uint32_t *items, uint32_t items_cap, uint32_t *first_item, uint64_t *launch_kmers,
                            uint32_t launches_cap, uint32_t *nitems, uint32_t *nlaunches)
{
	if (!off2 || !kpre2 || !items || !first_item || !launch_kmers || !nitems || !nlaunches)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_launches < 1 || limit == 0 || first_limit == 0)
		return fail(SDT_EINVAL, "bad argument (at least one launch, limits of at least one k-mer)");
	if (!sk_plan_count_items(off2, kpre2, nbuckets, first_limit, limit, max_launches, SK_COUNT_PACK_CHUNKS, SK_COUNT_ITEM_CHUNKS, items, items_cap,
	                         first_item, launch_kmers, launches_cap, nitems, nlaunches))
		return fail(SDT_ENOMEM, "output arrays too small");
	return SDT_OK;
}

int sdt_kmer_bucket(const uint64_t *key_words_msw_first, int K)
{
	// the level-1 minimizer bucket (0..255) of a canonical k-mer, as on the device
	if (!key_words_msw_first || K < 13 || K > 127)
		return -1;
	const int nw = K <= 31 ? 1 : (K <= 63 ? 2 : 4), m = sk_minimizer_len(K);
	uint32_t best = 0xFFFFFFFFu;
	for (int p = 0; p + m <= K; p++) {
		uint32_t fw = 0;
		for (int i = 0; i < m; i++) {
			const int bit = 2 * (K - 1 - (p + i));       // base p + i of the k-mer, counted from its low end
			const uint64_t w = key_words_msw_first[nw - 1 - bit / 64];
			fw = (fw << 2) | (uint32_t)((w >> (bit % 64)) & 3u);
		}
		const uint32_t hv = sk_mmer_hash(sk_canon_mmer(fw, m));
		if (hv < best) best = hv;
	}
	return (int)sk_l1_bucket(sk_bucket_hash(best));
}

int sdt_kmer_final_bucket(const uint64_t *key_words_msw_first, int K)
{
	// the final minimizer bucket (0 .. 2^18 - 1) of a k-mer: the unit of the count stage (csrc/sdt_minimizer.cuh,
	// the function the device's look-ups call)
	if (!key_words_msw_first || K < 13 || K > 127)
		return -1;
	if (K <= 31) { Key<1> k{{key_words_msw_first[0]}}; return (int)key_final_bucket<1>(k, K); }
	if (K <= 63) { Key<2> k{{key_words_msw_first[0], key_words_msw_first[1]}}; return (int)key_final_bucket<2>(k, K); }
	Key<4> k{{key_words_msw_first[0], key_words_msw_first[1], key_words_msw_first[2], key_words_msw_first[3]}};
	return (int)key_final_bucket<4>(k, K);
}

int sdt_gpu_table_info(sdt_ctx *c, uint64_t info[8])
{
	if (!c || !info)
		return fail(SDT_EINVAL, "NULL argument");
	for (int i = 0; i < 8; i++) info[i] = 0;
	info[1] = view_slots(c);
	info[2] = c->distinct_known;
	return SDT_OK;
}

int sdt_kmer_owner(const uint64_t *key_words_msw_first, int K, int nranks)
{
	// owner under EQUAL bucket ranges (what a context uses before its first sharded call has weighed the buckets)
	const int b = sdt_kmer_bucket(key_words_msw_first, K);
	if (b < 0 || nranks < 1)
		return -1;
	return sk_owner_of_bucket((uint32_t)b, nranks);
}

int sdt_gpu_keep_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads)
{
	if (!c || !packed_words || !offsets)
		return fail(SDT_EINVAL, "NULL argument");
	if (nreads == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++)
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	if (((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short");
	sdt_ctx::KeptBatch kb;
	kb.nwords = nwords; kb.nreads = nreads; kb.ord_base = c->ord_base; kb.ord_stride = c->ord_stride; kb.maxlen = maxlen;
	kb.d_words = (uint32_t *)keep_alloc(c, nwords * sizeof(uint32_t));
	kb.d_offs = (uint64_t *)keep_alloc(c, (nreads + 1) * sizeof(uint64_t));
	if (!kb.d_words || !kb.d_offs)
		return fail(SDT_ENOMEM, "kept reads: no device memory for another batch (%zu slabs held); run with --host-map", c->keep_slabs.size());
	c->kept.push_back(kb);
	HIPCHK(hipMemcpyAsync(kb.d_words, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipMemcpyAsync(kb.d_offs, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipStreamSynchronize(c->copy_stream));
	return SDT_OK;
}

int sdt_gpu_import_nodes(sdt_ctx *c, const uint64_t *keys, const uint32_t *l_links, const uint32_t *r_flags, const uint32_t *count,
                         const uint64_t *first, uint64_t n)
{
	if (!c || (n && (!keys || !l_links || !r_flags || !count)))
		return fail(SDT_EINVAL, "NULL argument");
	if (n == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	if ((double)(c->distinct_known + n) > (double)c->slots * MAX_LOAD) {
		rc = grow_table(c, c->distinct_known + n);
		if (rc != SDT_OK) return rc;
	}
	uint64_t *d_k = nullptr, *d_f = nullptr;
	uint32_t *d_l = nullptr, *d_r = nullptr, *d_c = nullptr;
	const uint64_t STEP = 1ULL << 26;                // nodes per upload: bounded staging memory
	const uint64_t m = n < STEP ? n : STEP;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	HIPCHK(hipMalloc((void **)&d_l, m * 4));
	HIPCHK(hipMalloc((void **)&d_r, m * 4));
	HIPCHK(hipMalloc((void **)&d_c, m * 4));
	if (first && c->d_first) HIPCHK(hipMalloc((void **)&d_f, m * 8));
	for (uint64_t i0 = 0; i0 < n && rc == SDT_OK; i0 += STEP) {
		const uint64_t k = n - i0 < STEP ? n - i0 : STEP;
		HIPCHK(hipMemcpyAsync(d_k, keys + i0 * c->nw, k * c->nw * 8, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_l, l_links + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_r, r_flags + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_c, count + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		if (d_f) HIPCHK(hipMemcpyAsync(d_f, first + i0, k * 8, hipMemcpyHostToDevice, c->stream));
		const int g = scan_grid(c, k);
		if (c->nw == 1) hipLaunchKernelGGL(k_import<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_import<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		else hipLaunchKernelGGL(k_import<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	(void)hipFree(d_k); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_c);
	if (d_f) (void)hipFree(d_f);
	return sync_stats(c);                            // a key that was already there shows up as SDT_EFULL
}

int sdt_gpu_count_reads_sharded(sdt_ctx *c, const void *d_packed_words, uint64_t nwords, const void *d_offsets, uint64_t nreads,
                                uint64_t max_read_len)
{
	(void)nwords;
	if (!c || (nreads && (!d_packed_words || !d_offsets)))
		return fail(SDT_EINVAL, "NULL argument");
	if (c->comm.kind == 0 || c->comm.nranks == 1) {
		if (nreads == 0)
			return SDT_OK;
		return sdt_gpu_count_reads_device(c, d_packed_words, nwords, d_offsets, nreads, max_read_len);
	}
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	Comm &cm = c->comm;
	sdt_ctx::SkState &k = c->sk;
	// agree on the geometry of the call: the longest read anywhere, the rank with the most reads
	std::vector<uint64_t> all((size_t)2 * cm.nranks);
	uint64_t mine[2] = {nreads, nreads ? max_read_len : 0};
	int rc = cm.allgather_host(mine, all.data(), sizeof mine);
	if (rc != SDT_OK) return rc;
	uint64_t maxlen = 0, maxreads = 0;
	for (int r = 0; r < cm.nranks; r++) {
		if (all[2 * r] > maxreads) maxreads = all[2 * r];
		if (all[2 * r + 1] > maxlen) maxlen = all[2 * r + 1];
	}
	if (maxreads == 0 || maxlen < (uint64_t)c->K + 1) {
		c->ord_base += nreads * c->ord_stride;
		return SDT_OK;
	}
	if (maxlen > (uint64_t)SK_MAX_READ_LEN || sk_geo(c->K, maxlen).smem > 160 * 1024)
		return fail(SDT_EINVAL, "reads of %llu bases do not fit the LDS tile of the sharded path", (unsigned long long)maxlen);
	if (c->ord_base + nreads * c->ord_stride >= SK_MAX_READ_ORDINAL)
		return fail(SDT_EINVAL, "read ordinals past 2^34 do not fit a super-k-mer record");
	const uint64_t per_read = maxlen - c->K + 1;
	uint64_t want = maxreads * per_read;
	if (want > (1ULL << 31)) want = 1ULL << 31;       // rounds of at most 2 G k-mers per rank: the exchange overlaps the next round
	if (getenv("SDT_SHARD_ROUND_KMERS"))             // (tests: many small rounds)
		want = strtoull(getenv("SDT_SHARD_ROUND_KMERS"), nullptr, 10);
	if (!k.ready || k.cap_kmers < want) {
		if (k.ready && !k.cap_is_max) { HIPCHK(hipStreamSynchronize(c->stream)); sk_free(c); }
		rc = sk_alloc(c, want, per_read);
		if (rc != SDT_OK) return rc;
	}
	rc = shard_alloc(c);
	if (rc != SDT_OK) return rc;
	if (!c->sh.have_ranges) {
		// Ownership.  Minimizer buckets are far from equal (a highly expressed transcript's minimizers are giants), so
		// equal ranges of buckets would leave the ranks unequal work.  Weigh the buckets on a sample -- the first 2^18
		// reads of every rank's slice through the level-1 scatter -- and cut the 256 buckets into contiguous ranges of
		// equal weight.  Every rank computes the same cut from the all-gathered counts; the sample's records are dropped.
		const uint64_t sample = nreads < (1ULL << 18) ? nreads : (1ULL << 18);
		if (sample) {
			rc = sk_scatter_launch(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, sample, maxlen, c->ord_base, false);
			if (rc != SDT_OK) return rc;
		}
		rc = sk_list1(c);
		if (rc != SDT_OK) return rc;
		std::vector<uint32_t> mat((size_t)cm.nranks * (SK_NB1 + 1));
		rc = cm.allgather_host(k.h_off1, mat.data(), (SK_NB1 + 1) * sizeof(uint32_t));
		if (rc != SDT_OK) return rc;
		shard_cut_ranges(mat.data(), cm.nranks, c->sh.ranges);
		c->sh.have_ranges = true;
		rc = sk_reset_pool1(c);
		if (rc != SDT_OK) return rc;
	}
	// every rank must cut its reads into the same number of rounds
	uint64_t capmine = k.cap_kmers;
	std::vector<uint64_t> caps(cm.nranks);
	rc = cm.allgather_host(&capmine, caps.data(), sizeof capmine);
	if (rc != SDT_OK) return rc;
	uint64_t cap = caps[0];
	for (int r = 1; r < cm.nranks; r++) if (caps[r] < cap) cap = caps[r];
	if (getenv("SDT_SHARD_ROUND_KMERS") && cap > want) cap = want;
	uint64_t per_round = cap / per_read / SK_TILE_READS * SK_TILE_READS;
	if (per_round < (uint64_t)SK_TILE_READS) per_round = SK_TILE_READS;
	const uint64_t rounds = (maxreads + per_round - 1) / per_round;
	for (uint64_t i = 0; i < rounds; i++) {
		const uint64_t r0 = i * per_round;
		const uint64_t nr = r0 < nreads ? (nreads - r0 < per_round ? nreads - r0 : per_round) : 0;
		if (nr) {
			rc = sk_scatter_launch(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets + r0, nr, maxlen, c->ord_base + r0 * c->ord_stride, false);
			if (rc != SDT_OK) return rc;
			c->sh.kmers_scattered += nr * per_read;
		}
		k.flushing = true;                           // sync_stats must not try to drain the pipeline on its own in here
		rc = sk_flush_sharded(c);
		k.flushing = false;
		if (rc != SDT_OK) return rc;
	}
	k.flushing = true;
	rc = shard_finish_pending(c);
	k.flushing = false;
	c->ord_base += nreads * c->ord_stride;
	return rc;
}

int sdt_gpu_push_reads_sharded(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads)
{
	if (!c || (nreads && (!packed_words || !offsets)))
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)i);
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	}
	if (nreads && ((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short");
	// staging buffers of the single-rank path (slot 0); the call is synchronous with respect to them -- a batch that an earlier
	// asynchronous push left staged there is launched first
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	uint32_t *dw = nullptr;
	uint64_t *dof = nullptr;
	if (nreads) {
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->cap_words[0] < nwords) {
			if (c->d_words[0]) HIPCHK(hipFree(c->d_words[0]));
			c->d_words[0] = nullptr; c->cap_words[0] = 0;
			HIPCHK(hipMalloc((void **)&c->d_words[0], nwords * sizeof(uint32_t)));
			c->cap_words[0] = nwords;
		}
		if (c->cap_offs[0] < nreads + 1) {
			if (c->d_offs[0]) HIPCHK(hipFree(c->d_offs[0]));
			c->d_offs[0] = nullptr; c->cap_offs[0] = 0;
			HIPCHK(hipMalloc((void **)&c->d_offs[0], (nreads + 1) * sizeof(uint64_t)));
			c->cap_offs[0] = nreads + 1;
		}
		dw = c->d_words[0]; dof = c->d_offs[0];
		HIPCHK(hipMemcpyAsync(dw, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(dof, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	}
	const int rc = sdt_gpu_count_reads_sharded(c, dw, nwords, dof, nreads, maxlen);
	if (rc == SDT_OK)
		HIPCHK(hipStreamSynchronize(c->stream));     // the staging buffers may be overwritten by the next call
	return rc;
}

#ifdef SDT_SK_L2_LOG
// debug builds only (not declared in include/sdt_gpu.h): point the level-2 scatter's slot log at a device buffer of `cap` words
extern "C" int sdt_gpu_debug_l2_log(sdt_ctx *c, void *d_buf, uint64_t cap)
{
	unsigned long long *p = (unsigned long long *)d_buf, cp = cap;
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_l2_log), &p, sizeof p));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_l2_log_cap), &cp, sizeof cp));
	return SDT_OK;
}
#endif

int sdt_gpu_stage_times(sdt_ctx *c, double ms[SDT_NSTAGES], uint64_t counters[SDT_NCOUNTERS])
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (ms) {
		for (int i = 0; i < SDT_NSTAGES; i++) ms[i] = 0;
		for (size_t i = 0; i < c->ev_used; i++) {
			float t = 0;
			HIPCHK(hipEventElapsedTime(&t, c->ev[i].a, c->ev[i].b));
			ms[c->ev[i].stage] += t;
		}
	}
	if (counters) {
		HIPCHK(hipMemcpy(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost));
		counters[0] = c->h_stats->sk_merges;
		counters[1] = c->h_stats->sk_spills;
		counters[2] = c->h_stats->sk_direct;
		counters[3] = c->h_stats->sk_gens;
		counters[4] = c->sk.st_chunks1;
		counters[5] = c->sk.st_chunks2;
		counters[6] = c->sk.st_flushes;
		counters[7] = c->sk.cap_kmers;
		for (int i = 0; i < 4; i++)
			counters[8 + i] = c->h_stats->sk_cyc[i];
		for (int i = 0; i < 4; i++)
			counters[12 + i] = c->h_stats->sk_cyc1[i];
		counters[16] = c->h_stats->sk_distinct_recs;
		counters[17] = c->h_stats->sk_records;
		counters[18] = c->h_stats->sk_distinct_kmers;
		counters[19] = 0;
	}
	return SDT_OK;
}

} // extern "C"

// ---- what the graph unit (sdt_gpu_graph.hip) sees of a context ----------------------------------------------------
sdti::GraphView sdti::graph_view(sdt_ctx *c)
{
	GraphView v;
	v.device = c->device; v.K = c->K; v.nw = c->nw; v.cu_count = c->cu_count;
	v.slots = view_slots(c);
	v.d_ent = c->d_ent; v.d_aux = c->d_aux; v.d_first = c->d_first;
	v.d_stats = c->d_stats; v.h_stats = c->h_stats;
	v.stream = c->stream; v.copy_stream = c->copy_stream;
	v.d_idx = &c->d_idx; v.idx_slots = &c->idx_slots; v.idx_n = &c->idx_n;
	v.gx = &c->gx;
	return v;
}

int sdti::sync_stats(sdt_ctx *c) { return ::sync_stats(c); }

// the first-occurrence ordinals have done their work (the device has the visiting order): 8 bytes per table slot go back to the arena
int sdti::drop_first(sdt_ctx *c)
{
	HIPCHK(hipSetDevice(c->device));
	if (c->d_first) { HIPCHK(hipFree(c->d_first)); c->d_first = nullptr; }
	return SDT_OK;
}

int sdti::release_pass1(sdt_ctx *c)
{
	const int rc = ::sync_stats(c);
	if (rc != SDT_OK) return rc;
	if (!getenv("SDT_KEEP_POOLS")) {                         // (measurement switch)
		sk_free(c);
	}
	return SDT_OK;
}

// pageable host memory <-> device in pieces through pinned staging buffers: a few threads copy between the caller's
// memory and the staging buffers while the copy engine moves the neighbouring pieces (a plain hipMemcpy of pageable memory
// runs at a third of the link)
// the staging buffers live as long as the process (pinning 64 MiB costs milliseconds: a large export makes dozens of transfers)
struct BigStage {
	static constexpr int NB = 4;
	static constexpr size_t CH = (size_t)32 << 20;
	void *pin[NB] = {};
	hipEvent_t done[NB] = {};
	bool ok = false;
	std::mutex mu;
	bool init()
	{
		if (ok) return true;
		for (int i = 0; i < NB; i++)
			if (hipHostMalloc(&pin[i], CH, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
				return false;
		ok = true;
		return true;
	}
};
static BigStage g_stage;

static void host_copy_mt(void *d, const void *s, size_t n)
{
	constexpr int T = 8;
	if (n < ((size_t)4 << 20)) { memcpy(d, s, n); return; }
	std::thread th[T];
	for (int t = 0; t < T; t++) {
		const size_t a0 = n * t / T, a1 = n * (t + 1) / T;
		th[t] = std::thread([=] { memcpy((char *)d + a0, (const char *)s + a0, a1 - a0); });
	}
	for (int t = 0; t < T; t++) th[t].join();
}

static int big_copy(hipStream_t copy_stream, void *dst, const void *src, size_t bytes, bool to_device)
{
	constexpr int NB = BigStage::NB;
	const size_t CH = BigStage::CH;
	if (bytes < CH) {
		if (hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
		    hipStreamSynchronize(copy_stream) != hipSuccess)
			return fail(SDT_EHIP, "copy of %zu bytes failed", bytes);
		return SDT_OK;
	}
	std::lock_guard<std::mutex> lock(g_stage.mu);
	if (!g_stage.init()) return fail(SDT_ENOMEM, "pinned staging for a %zu-byte transfer", bytes);
	void **pin = g_stage.pin;
	hipEvent_t *done = g_stage.done;
	int rc = SDT_OK;
	const size_t npieces = (bytes + CH - 1) / CH;
	if (to_device) {
		bool used[NB] = {};
		for (size_t k = 0; k < npieces && rc == SDT_OK; k++) {
			const int b = (int)(k % NB);
			const size_t off = k * CH, n = bytes - off < CH ? bytes - off : CH;
			if (used[b] && hipEventSynchronize(done[b]) != hipSuccess) { rc = fail(SDT_EHIP, "upload: event wait failed"); break; }
			host_copy_mt(pin[b], (const char *)src + off, n);
			if (hipMemcpyAsync((char *)dst + off, pin[b], n, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
			    hipEventRecord(done[b], copy_stream) != hipSuccess) { rc = fail(SDT_EHIP, "upload: copy failed"); break; }
			used[b] = true;
		}
	} else {
		// pieces k .. k + NB - 2 are on the link while the threads drain piece k - 1
		for (size_t k = 0; k < npieces + NB - 1 && rc == SDT_OK; k++) {
			if (k >= (size_t)(NB - 1)) {
				const size_t j = k - (NB - 1);
				const int b = (int)(j % NB);
				const size_t off = j * CH, n = bytes - off < CH ? bytes - off : CH;
				if (hipEventSynchronize(done[b]) != hipSuccess) { rc = fail(SDT_EHIP, "download: event wait failed"); break; }
				host_copy_mt((char *)dst + off, pin[b], n);
			}
			if (k < npieces) {
				const int b = (int)(k % NB);
				const size_t off = k * CH, n = bytes - off < CH ? bytes - off : CH;
				if (hipMemcpyAsync(pin[b], (const char *)src + off, n, hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
				    hipEventRecord(done[b], copy_stream) != hipSuccess) { rc = fail(SDT_EHIP, "download: copy failed"); break; }
			}
		}
	}
	if (hipStreamSynchronize(copy_stream) != hipSuccess && rc == SDT_OK) rc = fail(SDT_EHIP, "transfer: sync failed");
	return rc;
}

int sdti::h2d_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes) { return big_copy(copy_stream, dst, src, bytes, true); }
int sdti::d2h_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes) { return big_copy(copy_stream, dst, src, bytes, false); }
