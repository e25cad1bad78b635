// sdt_sk_scatter_seq.cuh -- chunk reservation of the super-k-mer pipeline and its level-1 scatter with one lane per read
// (design notes: sdt_superkmer.cuh).  Templates and inline device functions only: included by sdt_gpu.hip (through
// sdt_superkmer_kernels.cuh) and by the sdt_scatter_seq_*.hip translation units that instantiate the kernel for every
// window length (23 instantiations compile for minutes in one unit).
#pragma once
#include <hip/hip_runtime.h>
#include "sdt_kmer.cuh"
#include "sdt_table.cuh"
#include "sdt_superkmer.cuh"

using namespace sdt;

// Chunk ids come from the pool in blocks of SK_BLK per workgroup (s_blk = next id | end of block << 32): one global
// atomic per SK_BLK chunks.  (One atomicAdd per chunk on the single pool counter was measured to cap BOTH scatter
// kernels: same-address device atomics run at well under 1 G/s on MI355X.)
constexpr uint32_t SK_BLK = 128;
// The level-2 scatter takes larger blocks: a workgroup that runs out of ids stands still for the round trip of the fetching lane's global
// atomic (the lanes that need a chunk spin, everybody else waits at the round's barrier), 0.7 M times per step of the 200 M-read workload
// with blocks of 128.  What an item leaves of its last block is retired (sk_retire_block) and counted in the pool's size (sk_alloc).
#ifndef SDT_SK_BLK2
#define SDT_SK_BLK2 1024
#endif
constexpr uint32_t SK_BLK2 = SDT_SK_BLK2;
constexpr uint32_t SK_DEAD = 0xFFFFFFFFu;        // meta of a chunk id that was handed to a workgroup but never used

#ifdef SDT_SK_L2_LOG
extern __device__ unsigned long long *g_l2_log;
extern __device__ unsigned long long g_l2_log_cap;
// (debug build: every id sk_alloc_chunk hands out -- bit 63 | path << 61 | lane << 48 | end of block << 24 | id)
__device__ inline void sk_log_alloc(uint32_t path, uint32_t end, uint32_t id)
{
	if (g_l2_log) {
		const unsigned long long at = atomicAdd(g_l2_log, 1ULL);
		if (at + 1 < g_l2_log_cap)
			g_l2_log[1 + at] = (1ULL << 63) | ((unsigned long long)path << 61) | ((unsigned long long)(threadIdx.x & 0x1FFF) << 48) |
			                   ((unsigned long long)(end & 0xFFFFFFu) << 24) | (unsigned long long)(id & 0xFFFFFFu);
	}
}
// every value a lane reads from a chunk cursor: bit 63 | 3 << 61 | kind << 59 (0 slot granted, 1 this lane opens the next chunk,
// 2 look again, 3 the exchange it then wrote) | workgroup & 15 << 55 | lane << 45 | bucket << 35 | min(pos, 31) << 30 | chunk & 2^30 - 1
__device__ inline void sk_log_cursor(uint32_t kind, uint32_t lb, uint32_t chunk, uint32_t pos)
{
	if (g_l2_log) {
		const unsigned long long at = atomicAdd(g_l2_log, 1ULL);
		if (at + 1 < g_l2_log_cap)
			g_l2_log[1 + at] = (1ULL << 63) | (3ULL << 61) | ((unsigned long long)kind << 59) | ((unsigned long long)(blockIdx.x & 15u) << 55) |
			                   ((unsigned long long)(threadIdx.x & 1023u) << 45) | ((unsigned long long)(lb & 1023u) << 35) |
			                   ((unsigned long long)(pos > 31u ? 31u : pos) << 30) | (unsigned long long)(chunk & 0x3FFFFFFFu);
	}
}
#else
__device__ inline void sk_log_alloc(uint32_t, uint32_t, uint32_t) {}
__device__ inline void sk_log_cursor(uint32_t, uint32_t, uint32_t, uint32_t) {}
#endif

__device__ inline uint32_t sk_alloc_chunk(unsigned long long *s_blk, const SkPool &pool, uint32_t blk = SK_BLK)
{
	// Flag form on purpose: a lane that finds the block exhausted (id > end) spins until the lane that took id == end has
	// fetched the next block -- possibly a lane of ITS OWN wave.  With `for (;;) { ... return ...; }` the compiler is free
	// to move the fetch behind the loop (it sits on an exit path), and then the spinning lanes wait for ever for a lane that
	// waits for them to leave the loop (tools/lds_cursor_stress.hip hung exactly so).  The work below is inside the loop body.
	uint32_t got = 0;
	bool done = false;
	while (!done) {
		{   // a block that is being replaced (next id > end) is left alone until the fetching lane's exchange: see sk_reserve
			const unsigned long long w = __hip_atomic_load(s_blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if ((uint32_t)w > (uint32_t)(w >> 32)) {
				__builtin_amdgcn_s_sleep(1);
				continue;
			}
		}
		const unsigned long long v = atomicAdd(s_blk, 1ULL);
		const uint32_t id = (uint32_t)v, end = (uint32_t)(v >> 32);
		if (id < end) {
			got = id;
			done = true;
			sk_log_alloc(0, end, id);
		} else if (id == end) {
			const uint32_t base = atomicAdd(pool.next, blk);
			atomicExch(s_blk, ((unsigned long long)(base + blk) << 32) | (unsigned long long)(base + 1u));
			got = base;
			done = true;
			sk_log_alloc(1, end, base);
		}
		// id > end: another lane of this workgroup is fetching the next block -- look again (behind the peek above)
	}
	return got;
}

// n CONTIGUOUS chunk ids out of the workgroup's block (the level-2 scatter: a sub-bucket that takes many records in one round needs
// several chunks at once).  The lane whose request crosses the end of the block retires what is left of it, fetches a fresh block
// with room for its n ids and a whole block behind them, and takes its ids from the front.  Why not the pool's counter directly (rounds
// 2-5 did that for n > 1): a returning atomic on ONE global address costs ~15 ns per wave-instruction chip-wide; with the keeper lanes
// of round 6 spread over all sixteen waves of a workgroup, skewed data (every round a few sub-buckets need a run of chunks) put several
// such instructions per workgroup and round on that one counter: split 48 -> 78 ms at sigma = 2.5 (profiles/r6/ab_job9*).
// Flag form as above; the same protocol: a cursor past the end means "being replaced", look again.
__device__ inline uint32_t sk_alloc_chunks(unsigned long long *s_blk, const SkPool &pool, uint32_t n, uint32_t blk)
{
	uint32_t got = SK_NOCHUNK;
	bool done = false;
	while (!done) {
		{
			const unsigned long long w = __hip_atomic_load(s_blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if ((uint32_t)w > (uint32_t)(w >> 32)) {
				__builtin_amdgcn_s_sleep(1);
				continue;
			}
		}
		const unsigned long long v = atomicAdd(s_blk, (unsigned long long)n);
		const uint32_t id = (uint32_t)v, end = (uint32_t)(v >> 32);
		if (id <= end && id + n <= end) {
			got = id;
			done = true;
		} else if (id <= end) {
			for (uint32_t x = id; x < end && x < pool.chunks; x++)     // (the rest of the old block: fewer than n ids)
				pool.meta[x] = SK_DEAD;
			const uint32_t base = atomicAdd(pool.next, n + blk);
			atomicExch(s_blk, ((unsigned long long)(base + n + blk) << 32) | (unsigned long long)(base + n));
			got = base;
			done = true;
		}
		// id > end: another lane of this workgroup is fetching the next block -- look again (behind the peek above)
	}
	return got;
}

// the ids of a block that were never handed out must not look like chunks of an earlier batch
__device__ inline void sk_retire_block(unsigned long long blk, const SkPool &pool)
{
	const uint32_t next = (uint32_t)blk, end = (uint32_t)(blk >> 32);
	for (uint32_t id = next; id < end && id < pool.chunks; id++)
		pool.meta[id] = SK_DEAD;
}

// Reserve one record slot in the open chunk of local bucket `lb` (s_cur[lb] = chunk << 32 | records used).  The lane
// that takes the slot one past the end opens a new chunk (and counts it for its bucket: the counting sort of the
// chunk ids by bucket needs no pass of its own).  The counters s_cnt[lb] are the WORKGROUP's (LDS), added to the
// global ones once at its end: one global atomic per chunk on 256 counters (8 cache lines) ran at ~0.8 G/s and
// was what bounded the level-1 scatter.  false: the pool is exhausted (the caller takes its slow path).
__device__ inline bool sk_reserve(unsigned long long *s_cur, unsigned long long *s_blk, uint32_t lb, uint32_t meta_bucket,
                                  uint32_t cap, const SkPool &pool, uint32_t *s_cnt, uint32_t &chunk, uint32_t &pos)
{
	bool done = false, ok = false;                   // (flag form: see sk_alloc_chunk)
	while (!done) {
		// A chunk that is being replaced (records used > cap) is left alone: PLAIN read, short sleep, next iteration.  Sixteen waves
		// hammering one LDS word with ds_add_rtn_u64 while the opener's ds_wrxchg_rtn_b64 is on its way is the one condition under
		// which a chunk was seen handed out twice (1024-lane level-2 geometry, profiles/r3/l2_1024_lane_loss.md).  No inner wait
		// loop: the opener may be a lane of THIS wave, and it only gets to run when the waiting lanes come round the loop.
		if ((uint32_t)__hip_atomic_load(&s_cur[lb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > cap) {
			__builtin_amdgcn_s_sleep(1);
			continue;
		}
		const unsigned long long cur = atomicAdd(&s_cur[lb], 1ULL);
		pos = (uint32_t)cur;
		chunk = (uint32_t)(cur >> 32);
		if (pos < cap) {
			ok = chunk != SK_NOCHUNK;
			done = true;
			sk_log_cursor(0, lb, chunk, pos);            // (debug build)
		} else if (pos == cap) {
			sk_log_cursor(1, lb, chunk, pos);
			uint32_t id = sk_alloc_chunk(s_blk, pool);
			if (id >= pool.chunks) {
				id = SK_NOCHUNK;
			} else {
				pool.meta[id] = meta_bucket | (cap << 24);
				atomicAdd(&s_cnt[lb], 1u);
			}
			atomicExch(&s_cur[lb], ((unsigned long long)id << 32) | 1ULL);
			chunk = id;
			pos = 0;
			ok = id != SK_NOCHUNK;
			done = true;
			sk_log_cursor(3, lb, id, 1);
		} else {
#if SDT_SK_L2_LOG >= 2
			sk_log_cursor(2, lb, chunk, pos);            // (every look-again too: slows the spinning lanes enough to hide the failure)
#endif
			// pos > cap: another lane of this workgroup is replacing the chunk -- look again (the peek at the top of the loop keeps
			// this lane off the word until the exchange has happened)
		}
	}
	return ok;
}

// records are 8-byte aligned (24 / 40 / 56 bytes): 8-byte accesses, consecutive lanes still cover consecutive bytes
template <int RW> __device__ inline void sk_store_record(uint64_t *dst, const uint64_t (&rec)[RW])
{
#pragma unroll
	for (int i = 0; i < RW; i++)
		dst[i] = rec[i];
}
// a record into a slot of a LEVEL-2 chunk: the whole slot is written (a 32-byte slot of a 24-byte record: the fourth word as zero, in
// one 16-byte store with the third -- every byte of the group's 128-byte line gets written, no block is left half done)
template <int NW> __device__ inline void sk_store_record2(uint64_t *dst, const uint64_t (&rec)[SkFmt<NW>::REC_WORDS])
{
	if constexpr (SkFmt<NW>::REC2_STRIDE == 4 && SkFmt<NW>::REC_WORDS == 3) {
		ulonglong2 *d = reinterpret_cast<ulonglong2 *>(dst);
		d[0] = make_ulonglong2(rec[0], rec[1]);
		d[1] = make_ulonglong2(rec[2], 0ULL);
	} else {
		sk_store_record<SkFmt<NW>::REC_WORDS>(dst, rec);
	}
}
template <int RW> __device__ inline void sk_load_record(const uint64_t *src, uint64_t (&rec)[RW])
{
#pragma unroll
	for (int i = 0; i < RW; i++)
		rec[i] = src[i];
}


// ---- level 1, one lane per read (1-word keys with windows of 13 / 21 m-mers and reads of <= 128 k-mers: K = 23 / 31 on
// 100..150 bp; 2-word keys with a window of 53: K = 63 on reads of <= 256 k-mers) ------------------------------
// The strip kernel above spends two thirds of its time in the window minima: 64 lanes hash 64 m-mers and five
// shuffles later hold 64 - w window minima.  Here a lane walks ONE read base by base: the canonical m-mer rolls
// (one shift per strand), and the sliding minimum over W m-mers is the block decomposition of van Herk / Gil-Werman
// -- suffix minima of the block of W hashes behind, a running prefix minimum of the block ahead, min of the two --
// with both blocks in registers (W is a template parameter so that every index is static).  ~3 compares per window
// instead of a shuffle tree, every hash computed once, all 64 lanes on their own read.  Runs are noted in a
// per-lane LDS list (bucket << 6 | n - 1; the runs of a read follow one another) and cut out of the tile when the read is done.
constexpr int SK_SEQ_TILE = 256;                 // reads per tile = lanes per workgroup
constexpr int SK_SEQ_RUNCAP = 24;                // runs per read the list holds (more: emitted on the spot -- K=31 on 250 bp does that
                                                 //  for most reads and is still 3x the strip kernel: 50 -> 17 ms per 6.6 G k-mers)
constexpr int SK_SEQ_MAX_KMERS = 256;            // k-mers per read (the list entry has 8 bits for the position, 6 for n - 1)


// cut run [j0, j0 + n) of a read out of the tile and append the record to its level-1 bucket
template <int NW>
__device__ inline void sk_emit_run(const uint32_t *words, int rb_r, int len_r, int nk_r, int K, uint64_t read_ord, int j0, int n,
                                   uint32_t fb, unsigned long long *s_cur, unsigned long long *s_blk, const SkPool &pool,
                                   uint32_t *s_cnt, const Table<NW> &tbl, uint32_t &claimed, uint32_t &failed, uint32_t &done,
                                   uint32_t &emitted)
{
	constexpr int BW = SkFmt<NW>::BW, RW = SkFmt<NW>::REC_WORDS;
	const int hp = j0 > 0, hn = j0 + n < nk_r;
	const uint32_t l1 = fb >> SK_L2BITS, l2 = fb & (SK_NB2 - 1);
	uint32_t chunk, pos;
	if (sk_reserve(s_cur, s_blk, l1, l1, SK_CAP1, pool, s_cnt, chunk, pos)) {
		const int len = hp + n + K - 1 + hn, ps = rb_r + j0 - hp;
		uint64_t rec[RW];
		rec[0] = sk_header(read_ord, (uint32_t)j0, l2, n, hp, hn);
#pragma unroll
		for (int k = 0; k < BW; k++) {
			uint64_t wv = 0;
			if (32 * k < len) {
				wv = sk_stream_word(words, ps + 32 * k);
				const int keep = len - 32 * k;
				if (keep < 32)
					wv &= ~0ULL << (64 - 2 * keep);
			}
			rec[1 + k] = wv;
		}
		sk_store_record<RW>(pool.recs + ((size_t)chunk * SK_CAP1 + pos) * RW, rec);
		emitted += (uint32_t)n;
	} else if (!tbl.ent) {
		// no chunk left and no table to fall back on (a sharded context: this rank does not own every key; the caller gets SDT_EFULL)
		failed += (uint32_t)n;
	} else {
		// no chunk left: these k-mers take the direct path (put_kmerset, one atomic per occurrence)
		for (int jj = j0; jj < j0 + n; jj++) {
			uint32_t prev, next;
			const Key<NW> key = chop_record<NW>(words, rb_r, len_r, jj, K, prev, next);
			const uint64_t ord = tbl.first ? (read_ord << 16) | (uint64_t)jj : ORD_NONE;
			if (!table_put<NW>(tbl, key, prev, next, claimed, ord))
				failed++;
			done++;
		}
	}
}

template <int NW, int W>
__global__ __launch_bounds__(SK_SEQ_TILE, NW == 1 ? 4 : 2) void k_sk_scatter_reads_seq(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                                                      uint64_t nreads, int K, int m, int ncap, int max_tile_words, SkPool pool,
                                                                      unsigned long long *__restrict__ g_cursors, unsigned long long *__restrict__ g_blk,
                                                                      uint32_t *__restrict__ g_cnt, Table<NW> tbl, Stats *stats,
                                                                      uint64_t ord_base, uint64_t ord_stride)
{
	constexpr int TR = SK_SEQ_TILE;
	extern __shared__ uint32_t smem[];
	unsigned long long *s_cur = (unsigned long long *)smem;                               // SK_NB1
	uint32_t *s_rb = (uint32_t *)(s_cur + SK_NB1);                                        // TR + 2
	uint32_t *s_runs = s_rb + TR + 2;                                                     // TR * SK_SEQ_RUNCAP
	uint32_t *s_words = s_runs + TR * SK_SEQ_RUNCAP;                                      // LDS_LEAD + max_tile_words
	__shared__ unsigned long long s_blk;
	__shared__ uint32_t s_cnt[SK_NB1];               // chunks opened per bucket (see k_sk_scatter_reads)
	const int tid = threadIdx.x;
	for (int i = tid; i < SK_NB1; i += TR) {
		s_cur[i] = g_cursors[(size_t)blockIdx.x * SK_NB1 + i];
		s_cnt[i] = 0;
	}
	if (tid == 0)
		s_blk = g_blk[blockIdx.x];
	const uint64_t ntiles = (nreads + TR - 1) / TR;
	const uint32_t mmask = (1u << (2 * m)) - 1u;
	const int topsh = 2 * (m - 1);
	uint32_t claimed = 0, failed = 0, done = 0, emitted = 0;
#ifdef SDT_SK_TICKS
	unsigned long long cyc[4] = {0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define SK_TICK(i) do { t1 = wall_clock64(); cyc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define SK_TICK(i) do { } while (0)
#endif
	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const uint64_t r0 = tile * TR;
		const int nr = (int)((nreads - r0) < (uint64_t)TR ? (nreads - r0) : (uint64_t)TR);
		const uint64_t word0 = offs[r0] >> 4;
		int nwords = (int)(((offs[r0 + nr] + 15) >> 4) - word0) + TAIL_PAD;
		if (nwords > max_tile_words)
			nwords = max_tile_words;                     // cannot happen when max_read_len was honoured
		s_rb[tid] = (uint32_t)(offs[r0 + (tid < nr ? tid : nr)] - (word0 << 4));
		if (tid == 0)
			s_rb[TR] = (uint32_t)(offs[r0 + nr] - (word0 << 4));
		if (tid < LDS_LEAD)
			s_words[tid] = 0;
		for (int i = tid; i < nwords; i += TR)
			s_words[LDS_LEAD + i] = packed[word0 + i];
		__syncthreads();
		SK_TICK(0);
		const uint32_t *words = s_words + LDS_LEAD;
		const int rb_r = (int)s_rb[tid];
		const int len_r = tid < nr ? (int)s_rb[tid + 1] - rb_r : 0;
		const int nk_r = len_r >= K + 1 ? len_r - K + 1 : 0;                // prlHashReads.c:592
		const uint64_t read_ord = ord_base + (r0 + (uint64_t)tid) * ord_stride;
		uint32_t *runs = s_runs + tid * SK_SEQ_RUNCAP;
		int nrun = 0;
		if (nk_r > 0) {
			const int nhv = len_r - m + 1;               // m-mers of the read; window j covers m-mers [j, j + W)
			// rolling canonical m-mer: fw holds the last m - 1 bases, rc their reverse complement one base up
			uint32_t fw = sk_stream_mmer(words, rb_r, m) >> 2;
			uint32_t rc = (sk_rev2bit32(fw ^ 0xAAAAAAAAu) >> (32 - 2 * (m - 1))) << 2;
			int pb = rb_r + m - 1;                       // stream index of the next base to enter
			auto next_hv = [&]() -> uint32_t {
				const uint32_t b = (words[pb >> 4] >> (30 - 2 * (pb & 15))) & 3u;
				pb++;
				fw = ((fw << 2) | b) & mmask;
				rc = (rc >> 2) | ((b ^ 2u) << topsh);
				return sk_mmer_hash(fw < rc ? fw : rc);
			};
			int j0 = 0;
			uint32_t fb0 = 0, pfb = 0;
			auto window = [&](int j, uint32_t mn) {
				const uint32_t fb = sk_final_bucket(sk_bucket_hash(mn));
				if (j == 0) {
					fb0 = fb;
				} else if (j - j0 >= ncap || fb != pfb) {        // (a full record ends ncap k-mers after ITS start, not at a multiple of
					                                         //  ncap in the read: reads that cover the same stretch then cut the same
					                                         //  records out of it, and k_sk_count's dedupe folds them into one)
					if (nrun < SK_SEQ_RUNCAP)
						runs[nrun] = (fb0 << 6) | (uint32_t)(j - j0 - 1);
					else
						sk_emit_run<NW>(words, rb_r, len_r, nk_r, K, read_ord, j0, j - j0, fb0, s_cur, &s_blk, pool, s_cnt, tbl, claimed, failed, done, emitted);
					nrun++;
					j0 = j;
					fb0 = fb;
				}
				pfb = fb;
			};
			uint32_t h[W], nh[W];
#pragma clang loop unroll(full)
			for (int i = 0; i < W; i++)
				h[i] = next_hv();                        // nhv >= W: the read has at least one k-mer
#pragma clang loop unroll(full)
			for (int i = W - 2; i >= 0; i--)
				h[i] = h[i] < h[i + 1] ? h[i] : h[i + 1];
			int p = W;                                   // next m-mer position
			for (int jb = 0; jb < nk_r; jb += W) {
				window(jb, h[0]);
				uint32_t pre = 0xFFFFFFFFu;
#pragma clang loop unroll(full)
				for (int i = 1; i < W; i++) {
					const uint32_t x = p < nhv ? next_hv() : 0xFFFFFFFFu;
					p++;
					nh[i - 1] = x;
					pre = x < pre ? x : pre;
					if (jb + i < nk_r)
						window(jb + i, h[i] < pre ? h[i] : pre);
				}
				nh[W - 1] = p < nhv ? next_hv() : 0xFFFFFFFFu;
				p++;
				h[W - 1] = nh[W - 1];
#pragma clang loop unroll(full)
				for (int i = W - 2; i >= 0; i--)
					h[i] = nh[i] < h[i + 1] ? nh[i] : h[i + 1];
			}
			// the last run of the read
			if (nrun < SK_SEQ_RUNCAP)
				runs[nrun] = (fb0 << 6) | (uint32_t)(nk_r - j0 - 1);
			else
				sk_emit_run<NW>(words, rb_r, len_r, nk_r, K, read_ord, j0, nk_r - j0, fb0, s_cur, &s_blk, pool, s_cnt, tbl, claimed, failed, done, emitted);
			nrun++;
		}
		SK_TICK(1);
		const int nlist = nrun < SK_SEQ_RUNCAP ? nrun : SK_SEQ_RUNCAP;
		for (int k = 0, jr = 0; k < nlist; k++) {        // (the listed runs are the read's first ones, back to back from k-mer 0)
			const uint32_t e = runs[k];
			const int nr = (int)(e & 63u) + 1;
			sk_emit_run<NW>(words, rb_r, len_r, nk_r, K, read_ord, jr, nr, e >> 6, s_cur, &s_blk, pool, s_cnt, tbl, claimed, failed, done, emitted);
			jr += nr;
		}
		__syncthreads();                             // the tile buffers are reused
		SK_TICK(3);
	}
#undef SK_TICK
	for (int i = tid; i < SK_NB1; i += TR) {
		g_cursors[(size_t)blockIdx.x * SK_NB1 + i] = s_cur[i];
		if (s_cnt[i])
			atomicAdd(&g_cnt[i], s_cnt[i]);
	}
	if (tid == 0) {
		g_blk[blockIdx.x] = s_blk;
#ifdef SDT_SK_TICKS
		for (int i = 0; i < 4; i++)
			atomicAdd(&stats->sk_cyc1[i], cyc[i]);
#endif
	}
	if (done) {
		atomicAdd(&stats->kmers, (unsigned long long)done);
		atomicAdd(&stats->sk_direct, (unsigned long long)done);
	}
	if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
#pragma unroll
	for (int d = 32; d > 0; d >>= 1)
		emitted += __shfl_down(emitted, d);
	if ((tid & 63) == 0 && emitted)
		atomicAdd(&stats->sk_emitted, (unsigned long long)emitted);
}


// ---- host side: one launch (persistent workgroups: exactly the resident ones -- a second wave of them would run half empty)
struct SkSeqLaunch {
	const uint32_t *words;
	const uint64_t *offs;
	uint64_t nreads;
	int K, m, ncap, mtw;
	SkPool pool;
	unsigned long long *cursors, *blk;
	uint32_t *cnt;
	Stats *stats;
	uint64_t ord_base, ord_stride;
	unsigned max_wgs;
	int cu_count;
	hipStream_t stream;
};

template <int NW, int W> inline hipError_t sk_seq_launch_one(const SkSeqLaunch &a, const Table<NW> &tbl)
{
	const size_t smem = (size_t)SK_NB1 * 8 + (size_t)(SK_SEQ_TILE + 2) * 4 + (size_t)SK_SEQ_TILE * SK_SEQ_RUNCAP * 4 + (size_t)(LDS_LEAD + a.mtw) * 4;
	hipError_t e = hipFuncSetAttribute((const void *)k_sk_scatter_reads_seq<NW, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	if (e != hipSuccess) return e;
	int per_cu = 0;
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sk_scatter_reads_seq<NW, W>, SK_SEQ_TILE, smem);
	if (e != hipSuccess) return e;
	uint64_t res = (uint64_t)(per_cu > 0 ? per_cu : 1) * (uint64_t)a.cu_count;
	if (res > a.max_wgs) res = a.max_wgs;
	const uint64_t nt = (a.nreads + SK_SEQ_TILE - 1) / SK_SEQ_TILE;
	const unsigned grid = (unsigned)(nt < res ? nt : res);
	hipLaunchKernelGGL((k_sk_scatter_reads_seq<NW, W>), dim3(grid), dim3(SK_SEQ_TILE), smem, a.stream, a.words, a.offs, a.nreads, a.K, a.m, a.ncap,
	                   a.mtw, a.pool, a.cursors, a.blk, a.cnt, tbl, a.stats, a.ord_base, a.ord_stride);
	return hipGetLastError();
}

// sdt_scatter_seq_{a,b,c,d}.hip: every odd window of 1-word keys with K >= 17 (w = 9..21) and of 2-word keys (w = 23..53),
// i.e. every odd K from 17 to 63; hipErrorInvalidValue for a window that has no instantiation
hipError_t sk_seq_launch_nw1(int w, const SkSeqLaunch &a, const Table<1> &tbl);
hipError_t sk_seq_launch_nw2_lo(int w, const SkSeqLaunch &a, const Table<2> &tbl);    // w = 23..33
hipError_t sk_seq_launch_nw2_mid(int w, const SkSeqLaunch &a, const Table<2> &tbl);   // w = 35..43
hipError_t sk_seq_launch_nw2_hi(int w, const SkSeqLaunch &a, const Table<2> &tbl);    // w = 45..53
