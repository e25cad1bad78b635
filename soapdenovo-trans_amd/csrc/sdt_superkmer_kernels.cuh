// sdt_superkmer_kernels.cuh -- kernels of the super-k-mer pipeline (design notes: sdt_superkmer.cuh).
// Included by sdt_pipeline.hip and sdt_sharded.hip behind sdt_tile.cuh (stage_tile / TileView / tile_find_read) and sdt_kmer.cuh (chop_record).
#pragma once

// k_sk_count geometry: 1024 lanes, tiles of 512 records; two workgroups per CU wherever the LDS table allows it (their fill /
// dedupe / count / merge phases overlap)
#ifndef SDT_SK_PREFETCH
#define SDT_SK_PREFETCH 1      // one dword of every record of the next tile is loaded (and dropped) during phases A-C: L2 warm-up (0: off)
#endif
#ifndef SDT_SK_CLAIM_BELOW
#define SDT_SK_CLAIM_BELOW 0      // flush: keys counted at most this often send their slot claim along with the loads (0: never; 2 measured 5 % slower: profiles/r3)
#endif
#ifndef SDT_SK_LOAD16
#define SDT_SK_LOAD16 1           // owned flush of 1-word keys: one 16-byte load per entry instead of two 8-byte ones
#endif
#ifndef SDT_SK_CLAIM2_BELOW
#define SDT_SK_CLAIM2_BELOW 0     // the same for 2-word keys (their claim is KEY_LOCKED, published by one 16-byte store)
#endif
#ifndef SDT_SK_FLUSH_NUM
#define SDT_SK_FLUSH_NUM 4        // the LDS table is flushed between rounds once it is FLUSH_NUM / 8 full
#endif
#ifndef SDT_SK_NW2_TWO
#define SDT_SK_NW2_TWO 1        // 2-word keys: 1 = two workgroups per CU (64 registers; needs SDT_SK_SLOTS_NW2 <= 1280)
#endif
#ifndef SDT_SK_SEQ_FLUSH
#define SDT_SK_SEQ_FLUSH 1         // multi-word keys, ordinals: owned merges one slot at a time (registers)
#endif
#ifndef SDT_SK_CNT_TPB
#define SDT_SK_CNT_TPB 1024
#endif
#ifndef SDT_SK_SLOTS_TRACK
#define SDT_SK_SLOTS_TRACK 1536
#endif
#ifndef SDT_SK_SLOTS_NW2
#define SDT_SK_SLOTS_NW2 1280
#endif
#ifndef SDT_SK_SLOTS_NW1
#define SDT_SK_SLOTS_NW1 2048
#endif
template <int NW, bool TRACK> struct SkCntGeo {
	static constexpr int TPB = SDT_SK_CNT_TPB;
	static constexpr int WAVES_PER_SIMD = (NW == 1 || (NW == 2 && SDT_SK_NW2_TWO)) ? 2 * (TPB / 256) : TPB / 256;      // two workgroups per CU where the table allows
	// records per tile: the first half of the workgroup's lanes bring one each.  (1024 -- every lane brings one, a table of 1280 or 1024 slots
	// to keep two workgroups per CU -- was built in round 6: count 144.9 -> 182.3 / 211.3 ms at C3, the smaller table flushes far more often)
	static constexpr int TILE = 512;
	// LDS table entries: 8 B per key word + 20 B of counters (+ 8 B ordinal).  1-word keys: 2048 slots = 75 KB with the tile, two
	// workgroups per CU; with ordinals half the table keeps it at two
#ifdef SDT_SK_TEST_SLOTS
	// test build (libsdt_gpu_smalllds.so, tests/test_spills.py): an LDS table this small overflows in nearly every round, so that the
	// spill path -- memory-side atomics on a table the same workgroup also writes with plain stores -- runs all the time
	static constexpr int SLOTS = SDT_SK_TEST_SLOTS;
#else
	static constexpr int SLOTS = NW == 1 ? (TRACK ? SDT_SK_SLOTS_TRACK : SDT_SK_SLOTS_NW1) : (NW == 2 ? SDT_SK_SLOTS_NW2 : 2048);
#endif
	// (any size: 1-word keys with ordinals take 1536 slots, 36 B each -- the most that leaves two workgroups per CU)
	static constexpr int FLUSH_AT = SLOTS * SDT_SK_FLUSH_NUM / 8;                            // flush + clear between rounds past this load ...
	static constexpr int MAXFILL = SLOTS - 8;                             // ... a round counts 4 k-mers per slot left below this one
	static constexpr int MAXN = NW == 1 ? 32 : 64;                        // k-mers per record at most (sk_max_run)
	static constexpr int IDXN = TILE * MAXN / 16;                         // coarse index: one entry per 16 k-mers of a tile
	static constexpr int REP = 2 * TILE;                                  // slots of the tile's record-dedupe table
	// region shared by the dedupe table (phase B) and prefix / map / coarse index (phases C, D)
	static constexpr size_t REGION = (((size_t)(TILE + 2) * 4 + (size_t)TILE * 2 + (size_t)IDXN * 2) + 7) & ~(size_t)7;
	static_assert(((LDS_LEAD + TAIL_PAD) & 1) == 0, "the LDS table behind the tile's words is 8-byte aligned");
	static_assert(REGION >= (size_t)REP * 4, "the dedupe table aliases prefix + map + index");
};
// the staged form (k_sk_scatter_records_staged, round 5) has no cursor that lanes fight over -- the 1024-lane geometry that lost a chunk
// in the kernel above (profiles/r3/l2_1024_lane_loss.md) is safe there, and twice the records in flight per CU are worth 6 ms per step
// (65.2 / 65.8 -> 59.8 / 59.0 ms on the same box): the kernel is bound by its own phases (two barriers and a serial book-keeping
// section per tile), not by bytes
#ifndef SDT_SK_L2S_TPB
#define SDT_SK_L2S_TPB 1024
#endif
constexpr int SK_L2S_TPB = SDT_SK_L2S_TPB;
constexpr int SK_LIST2_FILL_SHIFT = SK_CAP2 == 16 ? 28 : 27;           // list2 entry = chunk id (28 bits) | (records in use - 1) << 28 (SK_CAP2 = 16: four bits)

#include "sdt_sk_scatter_seq.cuh"      // chunk reservation helpers + the one-lane-per-read level-1 scatter (own header: its
                                        // many instantiations are compiled in translation units of their own)

// ---- level 1: reads -> super-k-mer records in 256 buckets ----------------------------------------------------
template <int NW>
__global__ __launch_bounds__(TPB) void k_sk_scatter_reads(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                                          uint64_t nreads, int K, int m, int ncap, int max_tile_words,
                                                          int tile_smem_words, int hv_words, int hv2_words, int bits_words, SkPool pool,
                                                          unsigned long long *__restrict__ g_cursors, unsigned long long *__restrict__ g_blk,
                                                          uint32_t *__restrict__ g_cnt, Table<NW> tbl, Stats *stats,
                                                          uint64_t ord_base, uint64_t ord_stride)
{
	constexpr int BW = SkFmt<NW>::BW, RW = SkFmt<NW>::REC_WORDS;
	extern __shared__ uint32_t smem[];
	unsigned long long *s_cur = (unsigned long long *)(smem + tile_smem_words);           // SK_NB1 (tile_smem_words is even)
	uint32_t *s_hv = (uint32_t *)(s_cur + SK_NB1);                                        // hv_words (even)
	uint32_t *s_hv2 = s_hv + hv_words;                                                    // hv2_words (long windows only)
	unsigned long long *s_bits = (unsigned long long *)(s_hv2 + hv2_words);               // bits_words
	uint32_t *s_pc = (uint32_t *)(s_bits + bits_words);                                   // bits_words + 1
	__shared__ unsigned long long s_blk;
	__shared__ uint32_t s_cnt[SK_NB1];               // chunks opened per bucket (added to g_cnt once, at the end: 256 counters
	const int tid = threadIdx.x;                     //  in 8 cache lines take ~1 G same-line atomics/s, measured)
	for (int i = tid; i < SK_NB1; i += TPB) {
		s_cur[i] = g_cursors[(size_t)blockIdx.x * SK_NB1 + i];
		s_cnt[i] = 0;
	}
	if (tid == 0)
		s_blk = g_blk[blockIdx.x];
	const int w = K - m + 1;                         // m-mers per k-mer
	const uint64_t ntiles = (nreads + SK_TILE_READS - 1) / SK_TILE_READS;
	uint32_t claimed = 0, failed = 0, done = 0, emitted = 0;
#ifdef SDT_SK_TICKS
	unsigned long long cyc[4] = {0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define SK_TICK(i) do { t1 = wall_clock64(); cyc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define SK_TICK(i) do { } while (0)
#endif
	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		for (int i = tid; i < bits_words; i += TPB)  // the run-start bitmap (the barriers inside stage_tile order this)
			s_bits[i] = 0;
		const TileView tv = stage_tile(smem, max_tile_words, packed, offs, tile * SK_TILE_READS, nreads, K, SK_TILE_READS);
		const int npos = (int)tv.rb[tv.nr];
		SK_TICK(0);
		const bool strips = w <= 49;                 // a wave's 64 lanes hold at least 15 whole windows
		const uint32_t *bhs = s_hv;                  // bucket hash of the k-mer at every position (long windows: see below)
		const uint32_t nk = tv.nk, nkr = (nk + 63u) & ~63u;
		if (strips) {
			// One wave per read, strips of 64 - w k-mers.  Lane l of a strip hashes the canonical m-mer at its position; a
			// doubling min over shuffles gives min[l, l + P), two of those cover the window [l, l + w) of the k-mer that
			// starts there; its bucket hash goes to LDS (the emission reads it back) and is compared with the lane before:
			// a k-mer starts a run when it is the first of its read, when its bucket differs from its predecessor's, or
			// when the record is full.  Lane 0 carries the predecessor of the strip's first k-mer.  The run starts are
			// OR-ed into the tile's bitmap (zeroed below, before the barriers of stage_tile of the NEXT tile).
			const int valid = 64 - w, lane = tid & 63;
			int P = 1;
			while (2 * P <= w)
				P *= 2;
			for (int r = tid >> 6; r < tv.nr; r += TPB / 64) {
				const int nk_r = (int)(tv.pre[r + 1] - tv.pre[r]);
				const int rb_r = (int)tv.rb[r], len_r = (int)tv.rb[r + 1] - rb_r;
				for (int j0 = 0; j0 < nk_r; j0 += valid) {
					const int j = j0 + lane - 1, p = rb_r + j;
					uint32_t x = (j >= 0 && j + m <= len_r) ? sk_mmer_hash(sk_canon_mmer(sk_stream_mmer(tv.words, p, m), m)) : 0xFFFFFFFFu;
					for (int d = 1; d < P; d <<= 1) {
						const uint32_t y = __shfl_down(x, d);
						x = y < x ? y : x;
					}
					const uint32_t y = __shfl_down(x, w - P);
					x = y < x ? y : x;
					const uint32_t bh = sk_bucket_hash(x), pbh = __shfl_up(bh, 1);
					const bool kv = lane >= 1 && lane <= valid && j < nk_r;
					if (kv)
						s_hv[p] = bh;
					const bool start = kv && (j == 0 || (j & (ncap - 1)) == 0 || sk_final_bucket(bh) != sk_final_bucket(pbh));
					const unsigned long long mask = __ballot(start) >> 1;         // bit i: k-mer j0 + i of the read
					if (lane == 0 && mask) {
						const uint32_t q0 = tv.pre[r] + (uint32_t)j0, sh = q0 & 63u;
						atomicOr(&s_bits[q0 >> 6], mask << sh);
						if (sh && (mask >> (64u - sh)))
							atomicOr(&s_bits[(q0 >> 6) + 1], mask >> (64u - sh));
					}
				}
			}
		} else {
			// long windows (K > 59): hash of the canonical m-mer at every base position of the tile, then the window minima
			// of ALL positions by a sparse table in LDS -- min over [p, p + 2d) from the minima over [p, p + d) and
			// [p + d, p + 2d), log2(w) passes ping-ponging between two arrays; a window of w m-mers is two overlapping
			// spans of P = 2^floor(log2 w).  (Scanning the w - 2 shared hashes per k-mer, as this branch used to, made
			// the level-1 scatter 419 of the 949 ms of a K = 95 step.)  The last pass leaves the BUCKET hash of every
			// k-mer position in `bhs`, as the strips do in s_hv.
			for (int p = tid; p < npos; p += TPB)
				s_hv[p] = p + m <= npos ? sk_mmer_hash(sk_canon_mmer(sk_stream_mmer(tv.words, p, m), m)) : 0xFFFFFFFFu;
			__syncthreads();
			int P = 1;
			while (2 * P <= w)
				P *= 2;
			uint32_t *src = s_hv, *dst = s_hv2;
			for (int d = 1; d < P; d <<= 1) {
				for (int p = tid; p < npos; p += TPB) {
					const uint32_t a = src[p], b = p + d < npos ? src[p + d] : 0xFFFFFFFFu;
					dst[p] = b < a ? b : a;
				}
				__syncthreads();
				uint32_t *t = src; src = dst; dst = t;
			}
			for (int p = tid; p < npos; p += TPB) {
				const uint32_t a = src[p], b = p + w - P < npos ? src[p + w - P] : 0xFFFFFFFFu;
				dst[p] = sk_bucket_hash(b < a ? b : a);
			}
			__syncthreads();
			bhs = dst;
			for (uint32_t q = tid; q < nkr; q += TPB) {
				bool start = false;
				if (q < nk) {
					const int r = tile_find_read(tv.pre, q);
					const int j = (int)(q - tv.pre[r]);
					const int p = (int)tv.rb[r] + j;
					if (j == 0 || (j & (ncap - 1)) == 0) {    // ncap is a power of two
						start = true;
					} else {
						start = sk_final_bucket(bhs[p]) != sk_final_bucket(bhs[p - 1]);
					}
				}
				const unsigned long long mask = __ballot(start);
				if ((tid & 63) == 0)
					s_bits[q >> 6] = mask;
			}
		}
		__syncthreads();
		SK_TICK(1);
		// exclusive prefix of the popcounts: s_pc[i] = run starts before word i
		const int nw64 = (int)(nkr >> 6);
		if (tid < 64) {
			uint32_t run = 0;
			for (int base = 0; base < nw64; base += 64) {
				const int i = base + tid;
				const uint32_t c = i < nw64 ? (uint32_t)__popcll(s_bits[i]) : 0u;
				uint32_t x = c;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t y = __shfl_up(x, d);
					if (tid >= d)
						x += y;
				}
				if (i < nw64)
					s_pc[i] = run + x - c;
				run += __shfl(x, 63);
			}
			if (tid == 0)
				s_pc[nw64] = run;
		}
		__syncthreads();
		const uint32_t ns = nw64 ? s_pc[nw64] : 0u;
		SK_TICK(2);
		// one lane per run: cut the record out of the tile and append it to its bucket
		for (uint32_t i = tid; i < ns; i += TPB) {
			int lo = 0, hi = nw64;
			while (hi - lo > 1) {
				const int mid = (lo + hi) >> 1;
				if (s_pc[mid] <= i) lo = mid; else hi = mid;
			}
			unsigned long long x = s_bits[lo];
			for (uint32_t k = i - s_pc[lo]; k; k--)
				x &= x - 1;
			const uint32_t q = (uint32_t)lo * 64u + (uint32_t)(__ffsll((long long)x) - 1);
			const unsigned long long rest = x & (x - 1);
			uint32_t qn;
			if (rest) {
				qn = (uint32_t)lo * 64u + (uint32_t)(__ffsll((long long)rest) - 1);
			} else {
				int wd = lo + 1;
				while (wd < nw64 && s_bits[wd] == 0)
					wd++;
				qn = wd < nw64 ? (uint32_t)wd * 64u + (uint32_t)(__ffsll((long long)s_bits[wd]) - 1) : nk;
			}
			const int n = (int)(qn - q);
			const int r = tile_find_read(tv.pre, q);
			const int j = (int)(q - tv.pre[r]);
			const int nk_r = (int)(tv.pre[r + 1] - tv.pre[r]);
			const int p0 = (int)tv.rb[r] + j;
			const int hp = j > 0, hn = j + n < nk_r;
			const uint32_t bh = bhs[p0];
			const uint64_t read_ord = ord_base + (tile * SK_TILE_READS + (uint64_t)r) * ord_stride;
			uint32_t chunk, pos;
			if (sk_reserve(s_cur, &s_blk, sk_l1_bucket(bh), sk_l1_bucket(bh), SK_CAP1, pool, s_cnt, chunk, pos)) {
				const int len = hp + n + K - 1 + hn, ps = p0 - hp;
				uint64_t rec[RW];
				rec[0] = sk_header(read_ord, (uint32_t)j, sk_l2_bucket(bh), n, hp, hn);
#pragma unroll
				for (int k = 0; k < BW; k++) {
					uint64_t wv = 0;
					if (32 * k < len) {
						wv = sk_stream_word(tv.words, ps + 32 * k);
						const int keep = len - 32 * k;
						if (keep < 32)
							wv &= ~0ULL << (64 - 2 * keep);
					}
					rec[1 + k] = wv;
				}
				sk_store_record<RW>(pool.recs + ((size_t)chunk * SK_CAP1 + pos) * RW, rec);
				emitted += (uint32_t)n;
			} else if (!tbl.ent) {
				failed += (uint32_t)n;                   // (a sharded context has no table to fall back on: sk_emit_run)
			} else {
				// no chunk left: these k-mers take the direct path (put_kmerset, one atomic per occurrence)
				const int len_r = (int)(tv.rb[r + 1] - tv.rb[r]);
				for (int jj = j; jj < j + n; jj++) {
					uint32_t prev, next;
					const Key<NW> key = chop_record<NW>(tv.words, (int)tv.rb[r], len_r, jj, K, prev, next);
					const uint64_t ord = tbl.first ? (read_ord << 16) | (uint64_t)jj : ORD_NONE;
					if (!table_put<NW>(tbl, key, prev, next, claimed, ord))
						failed++;
					done++;
				}
			}
		}
		__syncthreads();                             // the tile buffers are reused
		SK_TICK(3);
	}
#undef SK_TICK
	for (int i = tid; i < SK_NB1; i += TPB) {
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

// every workgroup's open chunks: write the number of records they hold; retire the rest of its block of chunk ids
static __global__ __launch_bounds__(256) void k_sk_seal(const unsigned long long *__restrict__ cursors, uint32_t n, const unsigned long long *__restrict__ blk,
                                                 uint32_t nblk, SkPool pool, uint32_t cap)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const unsigned long long cur = cursors[i];
		const uint32_t chunk = (uint32_t)(cur >> 32), pos = (uint32_t)cur;
		if (chunk != SK_NOCHUNK)
			pool.meta[chunk] = (pool.meta[chunk] & 0xFFFFFFu) | ((pos < cap ? pos : cap) << 24);
	}
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nblk; i += gridDim.x * 256u)
		sk_retire_block(blk[i], pool);
}

static __global__ __launch_bounds__(256) void k_sk_init_cursors(unsigned long long *cursors, uint32_t n, uint32_t cap, unsigned long long *blk, uint32_t nblk)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
		cursors[i] = ((unsigned long long)SK_NOCHUNK << 32) | cap;           // "one past the end": the first record opens a chunk
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nblk; i += gridDim.x * 256u)
		blk[i] = 0;                                                          // empty block: the first chunk fetches one
}

// ---- chunk lists per bucket (counting sort of chunk ids by bucket) ---------------------------------------------
// exclusive scans over nb buckets by ONE workgroup of 1024: off[0..nb] of cnt, and (kmers != NULL) kpre[0..nb] of kmers
static __global__ __launch_bounds__(1024) void k_sk_scan(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ off, uint32_t *__restrict__ fillcur,
                                                  int nb, const unsigned long long *__restrict__ kmers, unsigned long long *__restrict__ kpre)
{
	__shared__ unsigned long long s_a[1024], s_b[1024];
	const int t = threadIdx.x, per = (nb + 1023) / 1024;
	const int i0 = t * per, i1 = i0 + per < nb ? i0 + per : nb;
	unsigned long long a = 0, b = 0;
	for (int i = i0; i < i1; i++) {
		a += cnt[i];
		if (kmers) b += kmers[i];
	}
	s_a[t] = a;
	s_b[t] = b;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const unsigned long long va = t >= d ? s_a[t - d] : 0, vb = t >= d ? s_b[t - d] : 0;
		__syncthreads();
		s_a[t] += va;
		s_b[t] += vb;
		__syncthreads();
	}
	a = t ? s_a[t - 1] : 0;
	b = t ? s_b[t - 1] : 0;
	for (int i = i0; i < i1; i++) {
		off[i] = (uint32_t)a;
		fillcur[i] = 0;
		a += cnt[i];
		if (kmers) {
			kpre[i] = b;
			b += kmers[i];
		}
	}
	if (t == 1023) {
		off[nb] = (uint32_t)s_a[1023];
		if (kmers) kpre[nb] = s_b[1023];
	}
}

static __global__ __launch_bounds__(256) void k_sk_chunk_place(SkPool pool, const uint32_t *__restrict__ off, uint32_t *__restrict__ fillcur,
                                                        uint32_t *__restrict__ list)
{
	const uint32_t used = *pool.next, n = used < pool.chunks ? used : pool.chunks;
	for (uint32_t c = blockIdx.x * 256u + threadIdx.x; c < n; c += gridDim.x * 256u) {
		const uint32_t mt = pool.meta[c];
		if (mt == SK_DEAD)
			continue;
		const uint32_t b = mt & 0xFFFFFFu;
		// level 2 only: the entry carries the chunk's fill (1..16) above the 27-bit chunk id, so that k_sk_count needs no
		// look at pool.meta between the chunk id and the record (one memory round trip and one live register less)
		list[off[b] + atomicAdd(&fillcur[b], 1u)] = c | (((mt >> 24) - 1u) << SK_LIST2_FILL_SHIFT);
	}
}

// the same for few buckets (level 1: 256): a workgroup counts its stretch of chunk ids per bucket in LDS and reserves a
// run per bucket with ONE global atomic (25 M atomics on 256 addresses were 5.6 ms per call)
static __global__ __launch_bounds__(256) void k_sk_chunk_place_few(SkPool pool, const uint32_t *__restrict__ off, uint32_t *__restrict__ fillcur,
                                                            uint32_t *__restrict__ list, int nb)
{
	__shared__ uint32_t s_cnt[1024], s_base[1024];
	constexpr uint32_t STRETCH = 256u * 16u;
	const uint32_t used = *pool.next, n = used < pool.chunks ? used : pool.chunks;
	for (uint32_t c0 = blockIdx.x * STRETCH; c0 < n; c0 += gridDim.x * STRETCH) {
		for (int i = threadIdx.x; i < nb; i += 256)
			s_cnt[i] = 0;
		__syncthreads();
		uint32_t mine[16], bucket[16];
#pragma unroll
		for (int t = 0; t < 16; t++) {
			const uint32_t c = c0 + (uint32_t)t * 256u + threadIdx.x;
			const uint32_t mt = c < n ? pool.meta[c] : SK_DEAD;
			bucket[t] = mt == SK_DEAD ? 0xFFFFFFFFu : (mt & 0xFFFFFFu);
			mine[t] = mt == SK_DEAD ? 0u : atomicAdd(&s_cnt[bucket[t]], 1u);
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nb; i += 256)
			s_base[i] = s_cnt[i] ? atomicAdd(&fillcur[i], s_cnt[i]) : 0u;
		__syncthreads();
#pragma unroll
		for (int t = 0; t < 16; t++)
			if (bucket[t] != 0xFFFFFFFFu)
				list[off[bucket[t]] + s_base[bucket[t]] + mine[t]] = c0 + (uint32_t)t * 256u + threadIdx.x;
		__syncthreads();
	}
}

// ---- level 2: one item = a run of chunks of ONE level-1 bucket, split into its 1024 sub-buckets ---------------

#ifdef SDT_SK_L2_LOG
// debug build (tools/l2_lost_chunk.py): every slot the level-2 scatter hands out goes into a side buffer
//   word 0: entries written; then per entry  chunk << 32 | pos << 16 | lane
__device__ unsigned long long *g_l2_log = nullptr;
__device__ unsigned long long g_l2_log_cap = 0;
#endif

// ---- level 2, staged (round 5; the only level-2 scatter since round 6) ---------------------------------------------------
// Rounds 2-4 stored every record where its sub-bucket's open chunk had room: 24..56 bytes at a time into one of 1024 open chunks
// per workgroup, so a 128-byte line was written by five or six stores sweeps apart, and many lines left L2 before they were full
// (WRITE_SIZE 94.6 GB for 56 GB of records, profiles/r4; that kernel, k_sk_scatter_records, is gone).  Here a record waits in LDS until its
// sub-bucket has a GROUP of S records (4 of 24 bytes, 2 of 40 / 56).  Per ROUND every lane brings D records (round 5: one; round 6: two -- what a
// round costs, two barriers and the book-keeping, is paid once for 2048 records):
//   1   every record takes a ticket of its sub-bucket (one LDS atomic; the tickets start at the records already waiting)
//   1.5 LANE i KEEPS THE BOOKS OF SUB-BUCKET i (round 6; round 5: the lane that drew the sub-bucket's first new ticket -- about 40 of a
//       wave's 64 lanes, in every wave, once per 1024 records): complete groups of this round, room in the open chunk, new chunks
//       (contiguous ids), the records that waited (they leave with the first group: stored here), the state of the next round -- no
//       cursor that several lanes fight over (sk_reserve's protocol is not used here at all)
//   2   a record of a complete group goes to its slot of the group in global memory -- the S records of a group are stored within one
//       round, by up to S lanes: 96 / 80 / 112 contiguous bytes that meet in L2 --, the others wait in the stage
// and at the end of the item the waiting records are written as a last, partial group.  Chunks fill from slot 0 up, so the
// count stage's lists (chunk id + fill) stay what they were.
#ifndef SDT_SK_L2S_S1
#define SDT_SK_L2S_S1 4           // 1-word keys: records per group of the staged level-2 scatter (4 x 24 B = 96 contiguous bytes per store phase)
#endif
#ifndef SDT_SK_L2S_WGS
#define SDT_SK_L2S_WGS 1          // workgroups per CU the staged level-2 scatter is compiled for (2: 64 registers per lane)
#endif
#ifndef SDT_SK_L2S_RPL
#define SDT_SK_L2S_RPL 2          // records per lane and round (1-word keys; 2-word keys: at most 2; 4-word keys: 1 -- a record is 56 bytes of registers).
                                  // C3: split 50.8 (1) -> 44.8 ms (2); before the runs of chunk ids came out of the workgroup's block (sk_alloc_chunks)
                                  // 2 and 4 were SLOWER (69 / 98 ms against 54.6): more records per round = more sub-buckets that need a run of chunks
#endif
template <int NW> struct SkL2Stage {
	static constexpr int S = NW == 1 ? SDT_SK_L2S_S1 : 2;            // records per group
	static constexpr int GPC = SK_CAP2 / S;                          // groups per chunk
	static constexpr int RPL = NW == 1 ? SDT_SK_L2S_RPL : (NW == 2 ? (SDT_SK_L2S_RPL < 2 ? SDT_SK_L2S_RPL : 2) : 1);      // records per lane and round
	static constexpr size_t SMEM = (size_t)SK_NB2 * S * SkFmt<NW>::REC_WORDS * 8;
};

template <int NW>
__global__ __launch_bounds__(SK_L2S_TPB, SDT_SK_L2S_WGS) void k_sk_scatter_records_staged(SkPool src, const uint32_t *__restrict__ list1,
                                                                         const SkItem *__restrict__ items, SkPool dst,
                                                                         uint32_t *__restrict__ g_cnt, unsigned long long *__restrict__ g_kmers, Stats *stats)
{
	constexpr int RW = SkFmt<NW>::REC_WORDS, RW2 = SkFmt<NW>::REC2_STRIDE, S = SkL2Stage<NW>::S, GPC = SkL2Stage<NW>::GPC;
	constexpr int CPT = SK_L2S_TPB / SK_CAP1;         // chunks per sweep
	constexpr int D = SkL2Stage<NW>::RPL;            // sweeps (records per lane) per round
	extern __shared__ unsigned long long s_stage[];  // SK_NB2 x S records
	__shared__ uint32_t s_cnt[SK_NB2];               // tickets of the running round (start: the records waiting in the stage)
	__shared__ uint32_t s_open[SK_NB2];              // open chunk of the sub-bucket (SK_NOCHUNK: none)
	__shared__ uint32_t s_pub_open[SK_NB2], s_pub_new[SK_NB2], s_pub_g[SK_NB2];     // what phase 2 of the round needs: the open chunk and the groups used in it
	                                                                               // before the round (g: used | groups << 8), the first new chunk
	__shared__ unsigned char s_fill[SK_NB2], s_used[SK_NB2];                       // records waiting; groups used in the open chunk
	__shared__ uint32_t s_kc[SK_NB2], s_cc[SK_NB2];  // k-mers and chunks per level-2 bucket of this item
	__shared__ unsigned long long s_blk;
	const SkItem it = items[blockIdx.x];
	const int tid = threadIdx.x;
	for (int i = tid; i < SK_NB2; i += SK_L2S_TPB) {
		s_cnt[i] = 0;
		s_open[i] = SK_NOCHUNK;
		s_fill[i] = 0;
		s_used[i] = 0;
		s_kc[i] = 0;
		s_cc[i] = 0;
	}
	if (tid == 0)
		s_blk = 0;
	__syncthreads();
	uint32_t failed = 0;
	// n chunk ids (contiguous) out of the workgroup's block of ids; SK_NOCHUNK: the pool is exhausted
	auto alloc = [&](uint32_t n) -> uint32_t {
		const uint32_t id = sk_alloc_chunks(&s_blk, dst, n, SK_BLK2);
		return id < dst.chunks && id + n <= dst.chunks ? id : SK_NOCHUNK;
	};
	// slot of record `t` (ticket) of a sub-bucket whose open chunk had `used` groups in use before: in the open chunk while it lasts,
	// then in the new chunks
	auto slot_of = [&](uint32_t open, uint32_t used, uint32_t first_new, uint32_t t) -> uint64_t * {
		const uint32_t a = (open == SK_NOCHUNK ? (uint32_t)GPC : used) + t / S;
		if (a < (uint32_t)GPC)
			return dst.recs + ((size_t)open * SK_CAP2 + a * S + t % S) * RW2;
		if (first_new == SK_NOCHUNK)
			return nullptr;
		return dst.recs + ((size_t)(first_new + (a - GPC) / GPC) * SK_CAP2 + ((a - GPC) % GPC) * S + t % S) * RW2;
	};
	// records and chunk ids are loaded a round ahead / two rounds ahead (registers: this kernel runs one workgroup per CU)
	const uint32_t slot = (uint32_t)tid % SK_CAP1, cfirst = it.c0 + (uint32_t)tid / SK_CAP1;
	uint32_t id_a[D], id_b[D], fill_a[D];
	uint64_t rec_a[D][RW];
#pragma unroll
	for (int d = 0; d < D; d++) {
		const uint32_t ca = cfirst + (uint32_t)d * CPT, cb = ca + (uint32_t)D * CPT;
		id_a[d] = ca < it.c1 ? list1[ca] : SK_NOCHUNK;
		id_b[d] = cb < it.c1 ? list1[cb] : SK_NOCHUNK;
	}
#pragma unroll
	for (int d = 0; d < D; d++) {
		fill_a[d] = 0;
#pragma unroll
		for (int i = 0; i < RW; i++)
			rec_a[d][i] = 0;
		if (id_a[d] != SK_NOCHUNK) {
			fill_a[d] = src.meta[id_a[d]] >> 24;
			sk_load_record<RW>(src.recs + ((size_t)id_a[d] * SK_CAP1 + slot) * RW, rec_a[d]);
		}
	}
	// (the rounds are uniform: every lane of the workgroup runs the same number of them, barriers included)
	const uint32_t nsweep = (it.c1 - it.c0 + CPT - 1) / CPT;
	for (uint32_t sw = 0; sw < nsweep; sw += (uint32_t)D) {
		uint64_t rec[D][RW];
		uint32_t b2[D], t[D];
		bool valid[D];
#pragma unroll
		for (int d = 0; d < D; d++) {
			const uint32_t ci = cfirst + (sw + (uint32_t)d) * CPT;
#pragma unroll
			for (int i = 0; i < RW; i++)
				rec[d][i] = rec_a[d][i];
			const uint32_t fill = ci < it.c1 ? fill_a[d] : 0u;
			id_a[d] = id_b[d];
			const uint32_t cn = ci + 2u * (uint32_t)D * CPT;
			id_b[d] = cn < it.c1 ? list1[cn] : SK_NOCHUNK;
			if (id_a[d] != SK_NOCHUNK) {
				fill_a[d] = src.meta[id_a[d]] >> 24;
				sk_load_record<RW>(src.recs + ((size_t)id_a[d] * SK_CAP1 + slot) * RW, rec_a[d]);
			}
			valid[d] = slot < fill;
			// ---- 1: tickets
			b2[d] = 0;
			t[d] = 0;
			if (valid[d]) {
				b2[d] = sk_hdr_l2(rec[d][0]);
				t[d] = atomicAdd(&s_cnt[b2[d]], 1u);
				atomicAdd(&s_kc[b2[d]], (uint32_t)sk_hdr_n(rec[d][0]));
			}
		}
		__syncthreads();
		// ---- 1.5: lane i keeps the books of sub-bucket i
		for (int i = tid; i < SK_NB2; i += SK_L2S_TPB) {
			const uint32_t f = s_fill[i], total = s_cnt[i];
			if (total == f)
				continue;                                // nobody came
			const uint32_t G = total / S, left = total - G * S;
			const uint32_t open = s_open[i], used = open == SK_NOCHUNK ? (uint32_t)GPC : s_used[i];
			const uint32_t room = (uint32_t)GPC - used;
			uint32_t first_new = SK_NOCHUNK;
			if (G > room) {
				const uint32_t nnew = (G - room + GPC - 1) / GPC;
				first_new = alloc(nnew);
				if (first_new != SK_NOCHUNK) {
					for (uint32_t j = 0; j < nnew; j++)
						dst.meta[first_new + j] = (it.b1 * SK_NB2 + (uint32_t)i) | ((uint32_t)SK_CAP2 << 24);
					s_cc[i] += nnew;
				}
			}
			s_pub_open[i] = open;
			s_pub_new[i] = first_new;
			s_pub_g[i] = used | (G << 8);
			if (G && f) {                                // the records that waited leave with the first group
				for (uint32_t j = 0; j < f; j++) {
					uint64_t old[RW];
#pragma unroll
					for (int w = 0; w < RW; w++)
						old[w] = s_stage[((size_t)i * S + j) * RW + w];
					uint64_t *p = slot_of(open, used, first_new, j);
					if (p) sk_store_record2<NW>(p, old);
					else failed++;
				}
			}
			// the next round's state
			if (G) {
				const uint32_t a_end = used + G;         // groups in use, counted from the open chunk's first
				if (a_end <= (uint32_t)GPC) {
					s_used[i] = (unsigned char)a_end;
				} else if (first_new != SK_NOCHUNK) {
					s_open[i] = first_new + (a_end - GPC - 1) / GPC;
					s_used[i] = (unsigned char)((a_end - GPC - 1) % GPC + 1);
				} else {
					s_open[i] = SK_NOCHUNK;              // (pool exhausted: the records of this round are counted as failed below)
					s_used[i] = 0;
				}
			}
			s_fill[i] = (unsigned char)left;
			s_cnt[i] = left;
		}
		__syncthreads();
		// ---- 2: complete groups to global memory, the rest waits
#pragma unroll
		for (int d = 0; d < D; d++) {
			if (!valid[d])
				continue;
			const uint32_t g = s_pub_g[b2[d]], G = g >> 8, used = g & 0xFFu;
			if (t[d] / S < G) {
				uint64_t *p = slot_of(s_pub_open[b2[d]], used, s_pub_new[b2[d]], t[d]);
				if (p) sk_store_record2<NW>(p, rec[d]);
				else failed++;
			} else {
				const uint32_t at = t[d] - G * S;
#pragma unroll
				for (int i = 0; i < RW; i++)
					s_stage[((size_t)b2[d] * S + at) * RW + i] = rec[d][i];
			}
		}
	}
	__syncthreads();
	// the records still waiting: a last, partial group; the open chunks' fills; the item's counts
	for (int i = tid; i < SK_NB2; i += SK_L2S_TPB) {
		const uint32_t f = s_fill[i];
		uint32_t open = s_open[i], used = open == SK_NOCHUNK ? (uint32_t)GPC : s_used[i];
		if (f) {
			if (used == (uint32_t)GPC) {
				open = alloc(1);
				used = 0;
				if (open != SK_NOCHUNK)
					s_cc[i]++;
			}
			if (open != SK_NOCHUNK) {
				for (uint32_t j = 0; j < f; j++) {
					uint64_t rec[RW];
#pragma unroll
					for (int w = 0; w < RW; w++)
						rec[w] = s_stage[((size_t)i * S + j) * RW + w];
					sk_store_record2<NW>(dst.recs + ((size_t)open * SK_CAP2 + used * S + j) * RW2, rec);
				}
			} else {
				failed += f;
			}
		}
		if (open != SK_NOCHUNK)
			dst.meta[open] = (it.b1 * SK_NB2 + (uint32_t)i) | ((used * S + f) << 24);
		if (s_kc[i])
			atomicAdd(&g_kmers[it.b1 * SK_NB2 + i], (unsigned long long)s_kc[i]);     // (64 bits: a hot bucket of a 2^33-k-mer batch)
		if (s_cc[i])
			atomicAdd(&g_cnt[it.b1 * SK_NB2 + i], s_cc[i]);
	}
	__syncthreads();
	{   // what is left of the workgroup's block of ids must not look like chunks of an earlier batch (sk_retire_block, on all lanes)
		const unsigned long long blk = s_blk;
		const uint32_t next = (uint32_t)blk, end = (uint32_t)(blk >> 32);
		for (uint32_t id = next + (uint32_t)tid; id < end && id < dst.chunks; id += SK_L2S_TPB)
			dst.meta[id] = SK_DEAD;
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

// ---- exchange: pack the level-1 chunks of every destination rank's buckets into one contiguous run ------------------
struct SkGatherPlan {
	uint32_t begin[64];          // first position in the chunk list of the piece that goes to rank p
	uint32_t pre[65];            // exclusive prefix of the piece lengths
	uint32_t dst0[64];           // first chunk of the piece in its destination buffer
	int n, self;                 // ranks; the piece of rank `self` goes straight into the receive buffer
};

template <int RW>
__global__ __launch_bounds__(256) void k_sk_gather(SkPool pool, const uint32_t *__restrict__ list, SkGatherPlan plan,
                                                   uint64_t *__restrict__ send, uint32_t *__restrict__ send_meta,
                                                   uint64_t *__restrict__ recv, uint32_t *__restrict__ recv_meta)
{
	constexpr uint32_t CW = SK_CAP1 * RW;            // 64-bit words per chunk
	const uint32_t total = plan.pre[plan.n], lane = threadIdx.x & 63u;
	for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < total; g += gridDim.x * 4u) {
		int p = 0;
		while (p + 1 < plan.n && plan.pre[p + 1] <= g)
			p++;
		const uint32_t i = g - plan.pre[p];
		const uint32_t chunk = list[plan.begin[p] + i], out = plan.dst0[p] + i;
		const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(pool.recs + (size_t)chunk * CW);
		ulonglong2 *d = reinterpret_cast<ulonglong2 *>((p == plan.self ? recv : send) + (size_t)out * CW);
		for (uint32_t w = lane; w < CW / 2; w += 64u)
			d[w] = s[w];
		if (lane == 0)
			(p == plan.self ? recv_meta : send_meta)[out] = pool.meta[chunk];
	}
}

static __global__ __launch_bounds__(256) void k_sk_iota(uint32_t *p, uint32_t n)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
		p[i] = i;
}

// ---- count: persistent workgroups, one final bucket at a time ---------------------------------------------------
// Per tile of 512 records the work is done in four phases (SQ counter passes of the round-2 kernel, profiles/r3/: 222 vector +
// 131 scalar instructions per k-mer occurrence, vector issue 70 % busy, the LDS array 33 % -- instructions, not LDS, were the
// wall, so the way down is to do the per-k-mer work less often):
//   A  records -> LDS
//   B  DEDUPE: a bucket holds windows around a handful of minimizers, and reads that cover the same stretch of a transcript cut
//      IDENTICAL records out of it (same bases, same length, same context flags).  Every record looks itself up in a small hash
//      table of record indices (one CAS; on a hit, a 16..48-byte compare against the representative's words in LDS): duplicates
//      add 1 to the representative's weight (and, with ordinals, atomicMin their header into the representative's: the low 18
//      header bits of identical records are identical) and drop out.
//   C  exclusive prefix of the k-mers of the DISTINCT records, compact map, coarse index
//   D  one lane per k-mer of a distinct record: cut, canonical, LDS table probe, and TWO LDS atomic adds of the record's weight.
// LDS node = key word(s) + ten 16-bit counters in five words: {L0,L1} {L2,L3} {Lnone,Rnone} {R3,R2} {R1,R0} -- an occurrence adds
// its weight to ONE left field (its prev code, or "none") and ONE right field, so count = sum of the left fields and nothing
// is read before it is added to.  The fields cannot overflow: the table is merged into the node table and cleared before the
// k-mers counted into it since the last clear pass 65535 (`since`), whatever the keys are.  Clamping at 63 when a node is merged
// is exactly the reference's saturating ++ (newhash.c:77-94).
template <int NW> __device__ inline uint32_t sk_fold_key(const Key<NW> &k)
{
	uint32_t x = 0;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		x = __builtin_rotateleft32(x, 7) ^ (uint32_t)k.w[i];
		x = __builtin_rotateleft32(x, 9) ^ (uint32_t)(k.w[i] >> 32);
	}
	return x;
}
// slot of a key in a table of SLOTS slots (any number): multiplicative hash, then the high word of hash x SLOTS
template <int NW, int SLOTS> __device__ inline uint32_t sk_lds_hash(const Key<NW> &k)
{
	const uint32_t h = sk_fold_key<NW>(k) * 0x9E3779B1u;
	if ((SLOTS & (SLOTS - 1)) == 0)
		return h >> (32 - __builtin_ctz(SLOTS));
	return __umulhi(h, (uint32_t)SLOTS);
}

// find-or-claim in the LDS table; -1: no room (full table or too many probes)
template <int NW, int SLOTS>
__device__ __forceinline__ int sk_lds_locate(unsigned long long *s_key, uint32_t *s_fill, const Key<NW> &key, uint32_t maxfill, int max_probe = 96)
{
	uint32_t s = sk_lds_hash<NW, SLOTS>(key);
	int found = -2;                                  // (flag form for the claimer of a multi-word key: see table_locate)
	for (int probe = 0; probe < max_probe && found == -2;) {
		uint64_t k0 = __hip_atomic_load(&s_key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (k0 == KEY_EMPTY) {
			if (__hip_atomic_load(s_fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= maxfill)
				return -1;
			const uint64_t old = atomicCAS(&s_key[s], (unsigned long long)KEY_EMPTY, (unsigned long long)(NW == 1 ? key.w[0] : KEY_LOCKED));
			if (old == KEY_EMPTY) {
				atomicAdd(s_fill, 1u);
				if (NW > 1) {
#pragma unroll
					for (int i = 1; i < NW; i++)
						s_key[i * SLOTS + s] = key.w[i];
					__hip_atomic_store(&s_key[s], key.w[0], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					found = (int)s;
					continue;
				}
				return (int)s;
			}
			k0 = old;
		}
		if (NW > 1 && k0 == KEY_LOCKED)
			continue;                                // the claimer is writing the low words: look again
		bool same = k0 == key.w[0];
		if (NW > 1 && same) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // (a system-scope fence here cost 20x: it invalidates L2)
#pragma unroll
			for (int i = 1; i < NW; i++)
				same = same && (__hip_atomic_load(&s_key[i * SLOTS + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == key.w[i]);     // (a volatile access stays a FLAT load: the address space is not inferred through it)
		}
		if (same)
			return (int)s;
		s = s + 1 == (uint32_t)SLOTS ? 0u : s + 1;
		probe++;
	}
	return found == -2 ? -1 : found;
}

// `w` occurrences with neighbour codes prev / next (0..3, 4 = none) into the five field words of slot s
__device__ inline void sk_lds_update(uint32_t *s_f, int s, uint32_t prev, uint32_t next, uint32_t w)
{
	atomicAdd(&s_f[5 * s + (prev >> 1)], w << ((prev & 1u) << 4));              // {L0,L1} {L2,L3} {Lnone,..}
	atomicAdd(&s_f[5 * s + 4 - (next >> 1)], w << (16u - ((next & 1u) << 4)));  // {R1,R0} {R3,R2} {..,Rnone}
}

// the node table's val layout (count low 16 | r_links | l_links, 6-bit fields clamped) of an LDS node
__device__ inline uint64_t sk_lds_val(const uint32_t *f)
{
	uint64_t v = 0;
	uint32_t cnt = f[2] & 0xFFFFu;
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const uint32_t l = (f[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu, r = (f[4 - (b >> 1)] >> (16 - (b & 1) * 16)) & 0xFFFFu;
		cnt += l;
		v |= (uint64_t)(l > 63u ? 63u : l) << (6 * b);
		v |= (uint64_t)(r > 63u ? 63u : r) << (24 + 6 * b);
	}
	return v | ((uint64_t)cnt << 48);      // (cnt <= SK_CNT_MAX_SINCE: 16 bits)
}

// `w` occurrences of one (key, prev, next) as a table_merge operand
__device__ inline uint64_t sk_weighted_val(uint32_t w, uint32_t prev, uint32_t next)
{
	const uint64_t c = w > 63u ? 63u : w;
	uint64_t v = (uint64_t)(w & 0xFFFFu) << 48;
	if (prev < 4u) v |= c << (6 * prev);
	if (next < 4u) v |= c << (24 + 6 * next);
	return v;
}

constexpr uint32_t SK_CNT_MAX_SINCE = 65535;     // k-mers counted into the LDS table between two clears at most (16-bit fields)

// A flush MERGES every LDS node into the node table -- without atomics when the workgroup is the only writer of the bucket's keys in
// the launch, by one saturating compare-and-swap otherwise --, and a k-mer that finds no LDS slot goes there directly.  (Round 5 carried
// a second form of this kernel that appended the LDS nodes to a node log, folded into a bucket-major table afterwards; the fold cost
// what these merges cost -- profiles/r5/README.md -- and it was removed in round 6.  This kernel was k_sk_count_flat in rounds 2-5.)
// (1-word keys without ordinals: 64 registers per lane, so that two workgroups of 16 waves share a CU.  8 waves per SIMD also
// means 78 usable SCALAR registers -- 800 per SIMD in granules of 16, 16 of every wave's reserved -- and this kernel keeps
// about ninety uniform values: the overflow lives in lanes of vector registers.  Raising the scalar budget by hand
// (amdgpu_waves_per_eu(4, 8) + amdgpu_num_vgpr(32) + amdgpu_num_sgpr(96)) removed every spill and cost the second workgroup
// per CU: 172 -> 251 ms per step on the 200 M-read workload.)
template <int NW, bool TRACK>
__global__ __launch_bounds__((SkCntGeo<NW, TRACK>::TPB), (SkCntGeo<NW, TRACK>::WAVES_PER_SIMD)) void k_sk_count(SkPool pool, const uint32_t *__restrict__ list2, const uint4 *__restrict__ items,
                                                         uint32_t item0, uint32_t item1, uint32_t *__restrict__ next_item, int K,
                                                         Table<NW> tbl, Stats *stats)
{
	using G = SkCntGeo<NW, TRACK>;
	constexpr int BW = SkFmt<NW>::BW, RW = SkFmt<NW>::REC_WORDS, SLOTS = G::SLOTS;
	constexpr uint32_t FLUSH_AT = G::FLUSH_AT, MAXFILL = G::MAXFILL;
	constexpr int SK_CNT_TPB = G::TPB;
	constexpr int TR = G::TILE;                      // records per tile: the first TR lanes bring one each
	constexpr int CPT = TR / SK_CAP2;                // chunks per tile
	constexpr int NWAVES = TR / 64;
	constexpr uint32_t REP = G::REP, REP_EMPTY = 0xFFFFFFFFu;
	extern __shared__ unsigned long long sm64[];
	// (the tile's small arrays first: every base below 64 KB is an immediate offset of a ds instruction, not a register)
	unsigned long long *s_h0 = sm64;                                     // TR: headers (TRACK: the smallest among a record's duplicates)
	uint32_t *s_w = (uint32_t *)(s_h0 + TR);                             // TR: weight of a distinct record
	uint32_t *s_pre = s_w + TR;                                          // TR + 2   } this region is the dedupe table s_rep
	unsigned short *s_map = (unsigned short *)(s_pre + TR + 2);          // TR       } (REP words) during phase B
	unsigned short *s_idx = s_map + TR;                                  // IDXN: distinct record of every 16th k-mer
	uint32_t *s_rep = s_pre;
	uint32_t *s_words = (uint32_t *)((char *)s_pre + G::REGION);         // LDS_LEAD + TR * BW * 2 + TAIL_PAD (an even number of words)
	unsigned long long *s_key = (unsigned long long *)(s_words + LDS_LEAD + TR * BW * 2 + TAIL_PAD);     // NW x SLOTS, word-major
	unsigned long long *s_ord = s_key + NW * SLOTS;                      // SLOTS when TRACK
	uint32_t *s_f = (uint32_t *)(s_ord + (TRACK ? SLOTS : 0));           // 5 x SLOTS
	__shared__ uint32_t s_fillc[2], s_item, s_spilled;       // s_fillc: keys in the LDS table = the sum of two counters, see phase D
	// statistics of the workgroup (claimed, failed, merges, spills, gens, k-mers, records, distinct records, their k-mers): in LDS, not
	// in nine registers per lane that live across every phase (the kernel has 64 registers: two workgroups of 16 waves per CU)
	enum { ST_CLAIMED, ST_FAILED, ST_MERGES, ST_SPILLS, ST_GENS, ST_KMERS, ST_RECS, ST_DRECS, ST_DKMERS, ST_N };
	__shared__ uint32_t s_stat[ST_N];
	__shared__ uint32_t s_ent[3 * (G::TILE / SK_CAP2)];      // ring of list entries: the tiles t, t + 1, t + 2 (see below)
	__shared__ unsigned long long s_wsum[NWAVES];
	const int tid = threadIdx.x;
	for (int i = tid; i < SLOTS; i += SK_CNT_TPB) {
		s_key[i] = KEY_EMPTY;
		if (TRACK) s_ord[i] = ORD_NONE;
	}
	for (int i = tid; i < 5 * SLOTS; i += SK_CNT_TPB)
		s_f[i] = 0;
	if (tid < LDS_LEAD)
		s_words[tid] = 0;
	if (tid < TAIL_PAD)
		s_words[LDS_LEAD + TR * BW * 2 + tid] = 0;
	if (tid == 0)
		s_fillc[0] = s_fillc[1] = 0;
	if (tid < ST_N)
		s_stat[tid] = 0;
	uint32_t *words = s_words + LDS_LEAD;
#ifdef SDT_SK_TICKS
	unsigned long long cyc[4] = {0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define SK_TICK(i) do { t1 = wall_clock64(); cyc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define SK_TICK(i) do { } while (0)
#endif
	// The owned flush of 1-word keys works on registers: every lane takes its PER slots out of LDS, issues all node-table loads, then
	// merges (flush_finish).  (An experiment that finished the merges a tile later -- to hide the round trip behind the next tile's
	// phases A-C -- measured no gain, the flushes are bound by HBM traffic, and was removed: profiles/r3/count_kernel_experiments.md.)
	constexpr int PER = (SLOTS + SK_CNT_TPB - 1) / SK_CNT_TPB;
	Key<NW> mk[PER];
	uint64_t madd[PER], mord[TRACK ? PER : 1];       // (the slot of a key is hashed again when the flush is finished: two registers
	bool have[PER];                                  //  per key less to carry across a tile)
	EntSnap<NW, TRACK> sn[PER];
	uint32_t ko = 0, km = 0;                         // (uniform) the two key counters as of the last barrier: the one that stands still in the coming round; the one the round adds to (bit 31: which)
	uint32_t room_shift = NW == 1 ? 2u : 1u;          // (uniform) a round of phase D takes 1, 2 or 4 k-mers per free slot of the LDS table
	bool stores_pending = false;                     // (uniform) plain stores of an owned flush may still be in flight
	auto flush_finish = [&]() {
		uint32_t claimed = 0, failed = 0, merges = 0;
#pragma unroll
		for (int p = 0; p < PER; p++)
			if (have[p]) {
				merges++;
				if (!table_merge_owned_at<NW, TRACK>(tbl, mk[p], flat_home<NW>(tbl, mk[p]), sn[p], madd[p], 0u, claimed, TRACK ? mord[TRACK ? p : 0] : ORD_NONE))
					failed++;
			}
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) {
			claimed += __shfl_down(claimed, d);
			failed += __shfl_down(failed, d);
			merges += __shfl_down(merges, d);
		}
		if ((tid & 63) == 0) {
			if (claimed) atomicAdd(&s_stat[ST_CLAIMED], claimed);
			if (failed) atomicAdd(&s_stat[ST_FAILED], failed);
			if (merges) atomicAdd(&s_stat[ST_MERGES], merges);
		}
		stores_pending = true;
	};
	// work items = runs of chunks of one bucket (a giant bucket is several items: every piece is counted and merged on
	// its own), handed out first come first served
	for (;;) {
		if (tid == 0) {
			s_item = item0 + atomicAdd(next_item, 1u);
			s_spilled = 0;
		}
		__syncthreads();
		const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);     // (uniform values belong in scalar registers)
		__syncthreads();
		if (item >= item1)
			break;
		const uint4 it = items[item];                // c0, c1 | whole (the bucket words: sdt_count_plan.h)
		const uint32_t ity = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.y);
		const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.x), c1 = ity & 0x7FFFFFFFu;
		const bool whole = (ity >> 31) != 0;        // the item is a whole bucket: nobody else touches its keys in this launch
		// Chunk id -> record are two dependent memory round trips per tile, and on this kernel's 64-register budget nothing can
		// wait in registers across the counting loop: the compiler spilled every such value (the next record, its chunk id, even
		// one prefetch dword), i.e. waited for the load at once -- with both trips in the open phases A-C were a third of the
		// kernel's time (tick counters, profiles/r3).  So the prefetches live across phases A-C only, where registers are free:
		// at the top of tile t wave 0 asks for the list entries of tile t + 2 and every record lane for ONE dword of its record
		// of tile t + 1 (which pulls the record's line into L2); at the end of phase C the entries go into a ring of three rows
		// in LDS and the dword is dropped.  The real record load at the top of a tile is then an L2 hit behind a known address.
		// A list entry carries the chunk's fill (k_sk_chunk_place): pool.meta is not read here.  (LDS-DMA -- global_load_lds_dword,
		// no destination register at all -- was tried for both and ran 10..50x slower than no prefetch: profiles/r3.)
		auto rec_ptr = [&](uint32_t e, uint32_t t) -> const uint64_t * {
			return pool.recs + ((size_t)(e & ((1u << SK_LIST2_FILL_SHIFT) - 1u)) * SK_CAP2 + t % SK_CAP2) * SkFmt<NW>::REC2_STRIDE;
		};
		auto rec_ok = [&](uint32_t e, uint32_t t) -> bool { return e != SK_NOCHUNK && t % SK_CAP2 <= (e >> SK_LIST2_FILL_SHIFT); };
		auto ent_row = [&](uint32_t t) -> uint32_t * { return s_ent + (t % 3u) * CPT; };
		if (tid < CPT) {
			ent_row(0)[tid] = c0 + (uint32_t)tid < c1 ? list2[c0 + tid] : SK_NOCHUNK;
			ent_row(1)[tid] = c0 + CPT + (uint32_t)tid < c1 ? list2[c0 + CPT + tid] : SK_NOCHUNK;
		}
		__syncthreads();
		uint32_t tile_no = 0;
		uint32_t since = 0;                          // k-mers counted into the LDS table since its last clear (uniform)
		SK_TICK(0);
		for (uint32_t cb = c0; cb < c1; cb += CPT) {
			// ---- A: the first TR lanes put their record into LDS
			// (phases A-C address everything from an opaque copy of the lane id: hoisted out of the tile loop, lane-dependent
			// addresses would sit in -- spilled -- registers, and every reload of a spilled register waits for ALL loads in
			// flight, the prefetches included)
			uint32_t ot_ = (uint32_t)tid;
			asm volatile("" : "+v"(ot_));
			const int ot = (int)ot_;
			uint32_t n = 0;
			uint64_t h0 = 0;
			uint64_t nx[RW];
			uint32_t ring_e = SK_NOCHUNK, pf = 0;        // prefetches: in flight during phases A-C
			const uint32_t e = ot < TR ? ent_row(tile_no)[ot / SK_CAP2] : SK_NOCHUNK;
			const uint32_t e1 = ot < TR ? ent_row(tile_no + 1)[ot / SK_CAP2] : SK_NOCHUNK;
			const bool ok = rec_ok(e, ot);
			// (every lane loads, from a harmless address when it has nothing to load: behind a branch the compiler cannot count
			// the loads in flight and waits for all of them; and the prefetches are issued BEHIND the loads this phase waits
			// for, because vector memory returns in order)
			sk_load_record<RW>(ok ? rec_ptr(e, ot) : pool.recs, nx);
			const bool ring_ok = ot < CPT && cb + 2 * CPT + (uint32_t)ot < c1;
			ring_e = (list2 + cb)[ring_ok ? 2 * CPT + ot : 0];
			if (!ring_ok)
				ring_e = SK_NOCHUNK;
			if (SDT_SK_PREFETCH)
				pf = *(const uint32_t *)(rec_ok(e1, ot) ? rec_ptr(e1, ot) : pool.recs);
			if (ot < TR) {
				if (ok) {
					h0 = nx[0];
					n = (uint32_t)sk_hdr_n(h0);
#pragma unroll
					for (int i = 0; i < BW; i++) {
						words[ot * BW * 2 + 2 * i] = (uint32_t)(nx[1 + i] >> 32);
						words[ot * BW * 2 + 2 * i + 1] = (uint32_t)nx[1 + i];
					}
				}
				s_h0[ot] = h0;
				s_w[ot] = 1;
			}
			for (uint32_t i = ot; i < REP; i += SK_CNT_TPB)
				s_rep[i] = REP_EMPTY;
			__syncthreads();                             // (also: the rounds of the last tile are over, the table is quiet)
			// ---- B: dedupe
			bool distinct = false;
			if (n) {
				distinct = true;
				uint32_t x = (uint32_t)h0 & SK_HDR_KIND_MASK;
#pragma unroll
				for (int i = 0; i < BW; i++) {
					x = __builtin_rotateleft32(x, 5) ^ (uint32_t)nx[1 + i];
					x = __builtin_rotateleft32(x, 11) ^ (uint32_t)(nx[1 + i] >> 32);
				}
				static_assert((REP & (REP - 1)) == 0, "the dedupe slot is the top bits of the hash");
				uint32_t slot = (x * 0x85EBCA77u) >> (32 - __builtin_ctz(REP));
				for (;;) {
					const uint32_t cur = atomicCAS(&s_rep[slot], REP_EMPTY, (uint32_t)ot);
					if (cur == REP_EMPTY)
						break;                           // this record represents its kind
					// identical records: same bases, and the same low 18 header bits (bucket, n, context flags)
					bool same = (((uint32_t)s_h0[cur] ^ (uint32_t)h0) & SK_HDR_KIND_MASK) == 0;
#pragma unroll
					for (int i = 0; i < BW; i++)
						same = same && (((uint64_t)words[cur * BW * 2 + 2 * i] << 32) | words[cur * BW * 2 + 2 * i + 1]) == nx[1 + i];
					if (same) {
						atomicAdd(&s_w[cur], 1u);
						if (TRACK)
							atomicMin(&s_h0[cur], (unsigned long long)h0);
						distinct = false;
						break;
					}
					slot = (slot + 1) & (REP - 1);
				}
			}
			// exclusive prefix sums over the distinct records: k-mers [15:0], records [31:16]; all k-mers of the tile [47:32], all its records [63:48]
			const uint32_t n_mine = distinct ? n : 0u;
			unsigned long long xs = ((unsigned long long)(n ? (n | 0x10000u) : 0u) << 32) | (distinct ? (n | 0x10000u) : 0u);
			const unsigned long long xs_mine = xs;
			if (ot < TR) {
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const unsigned long long y = __shfl_up(xs, d);
					if ((ot & 63) >= d)
						xs += y;
				}
				if ((ot & 63) == 63)
					s_wsum[ot >> 6] = xs;
			}
			__syncthreads();
			// ---- C: compact map of the distinct records
			unsigned long long wbase = 0, tot = 0;
#pragma unroll
			for (int wv = 0; wv < NWAVES; wv++) {
				const unsigned long long v = s_wsum[wv];
				if (wv < (ot >> 6)) wbase += v;
				tot += v;
			}
			const uint32_t tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tot), thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tot >> 32));
			const uint32_t total = tlo & 0xFFFFu, ndist = tlo >> 16, tile_kmers = thi & 0xFFFFu;
			// the 16-bit fields hold what this tile can add only if the table is young enough (uniform decision)
			bool want_flush = since + tile_kmers > SK_CNT_MAX_SINCE;
			since += tile_kmers;
			if (distinct) {
				const uint32_t ex = (uint32_t)(wbase + xs - xs_mine);
				const uint32_t lo = ex & 0xFFFFu, ci = (ex >> 16) & 0xFFFFu, hi = lo + n_mine;
				s_pre[ci] = lo;
				s_map[ci] = (unsigned short)ot;
				// coarse index: every 16th k-mer of the tile lies in exactly one record, which writes itself there
				// (the look-up below starts from it instead of searching the whole prefix array)
				for (uint32_t m16 = (lo + 15u) & ~15u; m16 < hi; m16 += 16u)
					s_idx[m16 >> 4] = (unsigned short)ci;
			}
			if (ot < CPT)
				ent_row(tile_no + 2)[ot] = ring_e;
			asm volatile("" :: "v"(pf));                 // (the warm-up dword dies here)
			if (ot == 0) {
				s_pre[ndist] = total;
				s_stat[ST_KMERS] += tile_kmers;
				s_stat[ST_RECS] += thi >> 16;
				s_stat[ST_DRECS] += ndist;
				s_stat[ST_DKMERS] += total;
			}
			__syncthreads();
			SK_TICK(1);
			// ---- D: rounds of up to 4 k-mers per free slot: barriers are what this loop pays for (a k-mer that does find the table
			// full takes the direct path)
			const bool last_tile = cb + CPT >= c1;
			tile_no++;
			for (uint32_t qb = 0;;) {
				if (want_flush) {
					SK_TICK(2);
					// merge every LDS node into the node table and clear it: plain read-modify-write when this workgroup is the
					// only writer of the bucket's keys, one saturating CAS per distinct key otherwise.  (The ONE place where it is
					// done -- before a tile that could overflow the fields, between rounds when the table is half full, after the
					// item's last round: three copies of the merge code cost the hot loop its registers.)
					const bool owned = whole && __builtin_amdgcn_readfirstlane((int)s_spilled) == 0;
					uint32_t claimed = 0, failed = 0, merges = 0;
					// (the stores of the previous flush were left in flight: they must have landed before this one reads)
					if (stores_pending) {
						asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						__syncthreads();
						stores_pending = false;
					}
					if (owned && (NW > 1 || TRACK) && SDT_SK_SEQ_FLUSH) {
						// multi-word keys, keys with ordinals: one slot at a time -- two keys, two snapshots and two addresses in flight did not fit the 64
						// registers, and a spilled snapshot is a load that is waited for at once (see the 1-word path below)
#pragma unroll 1
						for (int i = tid; i < SLOTS; i += SK_CNT_TPB) {
							if (s_key[i] == KEY_EMPTY)
								continue;
							Key<NW> key;
							key.w[0] = s_key[i];
#pragma unroll
							for (int wv = 1; wv < NW; wv++)
								key.w[wv] = s_key[wv * SLOTS + i];
							const uint64_t add = sk_lds_val(&s_f[5 * i]);
							const uint64_t ord = TRACK ? (uint64_t)s_ord[i] : ORD_NONE;
							s_key[i] = KEY_EMPTY;
#pragma unroll
							for (int f = 0; f < 5; f++)
								s_f[5 * i + f] = 0;
							if (TRACK) s_ord[i] = ORD_NONE;
							const uint64_t slot = flat_home<NW>(tbl, key);
							merges++;
							if (!table_merge_owned_at<NW, TRACK>(tbl, key, slot, ent_load<NW, TRACK>(tbl, slot, key, NW == 2 && SDT_SK_CLAIM2_BELOW > 0 && (add >> 48) <= SDT_SK_CLAIM2_BELOW), add, 0u, claimed, ord))
								failed++;
						}
#pragma unroll
						for (int d = 32; d > 0; d >>= 1) {
							claimed += __shfl_down(claimed, d);
							failed += __shfl_down(failed, d);
							merges += __shfl_down(merges, d);
						}
						if ((tid & 63) == 0) {
							if (claimed) atomicAdd(&s_stat[ST_CLAIMED], claimed);
							if (failed) atomicAdd(&s_stat[ST_FAILED], failed);
							if (merges) atomicAdd(&s_stat[ST_MERGES], merges);
						}
						stores_pending = true;
					} else if (owned) {
						// this lane's PER slots: everything out of LDS, all global loads issued, then (now or a tile later) the merges
#pragma unroll
						for (int p = 0; p < PER; p++) {
							const int i = tid + p * SK_CNT_TPB;
							have[p] = i < SLOTS && s_key[i] != KEY_EMPTY;
							if (have[p]) {
								mk[p].w[0] = s_key[i];
#pragma unroll
								for (int wv = 1; wv < NW; wv++)
									mk[p].w[wv] = s_key[wv * SLOTS + i];
								madd[p] = sk_lds_val(&s_f[5 * i]);
								if (TRACK) mord[p] = (uint64_t)s_ord[i];
								s_key[i] = KEY_EMPTY;
#pragma unroll
								for (int f = 0; f < 5; f++)
									s_f[5 * i + f] = 0;
								if (TRACK) s_ord[i] = ORD_NONE;
							}
						}
						__builtin_amdgcn_sched_barrier(0);       // (hashes first, then all loads, then the merges: interleaved by the
						if (NW == 1 && SDT_SK_LOAD16) {          //  scheduler the snapshots were spilled, i.e. waited for one by one)
							// key and val of a 16-byte entry with ONE agent-scope load (two 8-byte ones are two requests to the memory side)
							typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
							u32x4 raw[PER];
#pragma unroll
							for (int p = 0; p < PER; p++) {
								const Entry<NW> *e = tbl.ent + (have[p] ? flat_home<NW>(tbl, mk[p]) : 0);
								asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(raw[p]) : "v"(e) : "memory");
							}
							asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]) :: "memory");
#pragma unroll
							for (int p = 1; p < PER; p++)
								asm volatile("" : "+v"(raw[p]) :: "memory");
#pragma unroll
							for (int p = 0; p < PER; p++) {
								sn[p].k[0] = ((uint64_t)raw[p].y << 32) | raw[p].x;
								sn[p].v = ((uint64_t)raw[p].w << 32) | raw[p].z;
								sn[p].f = ORD_NONE;
								sn[p].won = false;
							}
						} else {
#pragma unroll
						for (int p = 0; p < PER; p++)
							if (have[p])       // (seen once or twice in this generation: an error k-mer, most likely new to the node table)
								sn[p] = ent_load<NW, TRACK>(tbl, flat_home<NW>(tbl, mk[p]), mk[p], SDT_SK_CLAIM_BELOW > 0 && (madd[p] >> 48) <= SDT_SK_CLAIM_BELOW);
						}
						__builtin_amdgcn_sched_barrier(0);
						flush_finish();
					} else {
						for (int i = tid; i < SLOTS; i += SK_CNT_TPB) {
							const uint64_t k0 = s_key[i];
							if (k0 == KEY_EMPTY)
								continue;
							Key<NW> key;
							key.w[0] = k0;
#pragma unroll
							for (int wv = 1; wv < NW; wv++)
								key.w[wv] = s_key[wv * SLOTS + i];
							merges++;
							const uint64_t add = sk_lds_val(&s_f[5 * i]);
							const uint64_t ord = TRACK ? (uint64_t)s_ord[i] : ORD_NONE;
							if (!table_merge<NW>(tbl, key, add, 0u, claimed, ord))
								failed++;
							s_key[i] = KEY_EMPTY;
#pragma unroll
							for (int f = 0; f < 5; f++)
								s_f[5 * i + f] = 0;
							if (TRACK) s_ord[i] = ORD_NONE;
						}
#pragma unroll
						for (int d = 32; d > 0; d >>= 1) {
							claimed += __shfl_down(claimed, d);
							failed += __shfl_down(failed, d);
							merges += __shfl_down(merges, d);
						}
						if ((tid & 63) == 0) {
							if (claimed) atomicAdd(&s_stat[ST_CLAIMED], claimed);
							if (failed) atomicAdd(&s_stat[ST_FAILED], failed);
#ifdef SDT_SK_SPLIT_MERGE_STAT                       // (measurement build: the merges by compare-and-swap are reported as `lds_spills`)
							if (merges) atomicAdd(&s_stat[ST_SPILLS], merges);
#else
							if (merges) atomicAdd(&s_stat[ST_MERGES], merges);
#endif
						}
					}
					__syncthreads();
					if (tid == 0) {
						s_fillc[0] = s_fillc[1] = 0;
						if (!(last_tile && qb >= total)) s_stat[ST_GENS]++;
					}
					__syncthreads();
					ko = 0;
					km &= 0x80000000u;
					since = tile_kmers;                  // (what is left of this tile is at most the tile)
					SK_TICK(3);
				}
				if (qb >= total)
					break;
				// The number of keys in the table decides how long a round is and when to flush, so every wave must see the SAME number:
				// a wave that read one live counter a little late -- after a faster wave had claimed the next round's first slots --
				// would take another branch than its workgroup and meet it at the wrong barrier (seen as hangs and lost k-mers once
				// merges ran between the barrier and the read).  Hence two counters: round r adds to counter r & 1 only, so the
				// other one stands still for the whole round and can be read at leisure; ko / km are the values all waves agree on.
				const uint32_t fill0 = ko + (km & 0x7FFFFFFFu);  // < FLUSH_AT here
				// The round behind an owned flush -- its plain stores may still be on their way -- must not send a k-mer to the node table
				// with a memory-side atomic: nothing orders another wave's plain store before that atomic on the same node.  So this one
				// round (the table is empty: fill0 = 0) takes no more k-mers than three quarters of the slots and probes without a bound:
				// every k-mer finds or claims a slot, whatever the keys are.  At the barrier that ends it every wave has waited for its own
				// stores; any spill of a later round is behind all of them.
				const bool guard = stores_pending;               // (uniform)
				const uint32_t room = guard ? (uint32_t)(SLOTS * 3 / 4) : (MAXFILL - fill0) << room_shift;
				const int max_probe = guard ? SLOTS : 96;
				uint32_t *const fillc = &s_fillc[km >> 31];
				const uint32_t maxfill = MAXFILL - ko;
				const uint32_t qe = qb + room < total ? qb + room : total;
				for (uint32_t q = qb + tid; q < qe; q += SK_CNT_TPB) {
					uint32_t ci = s_idx[q >> 4];             // the record of k-mer q & ~15; q's own is at most a few records on
					while (s_pre[ci + 1] <= q)
						ci++;
					const int r = s_map[ci];
					const int j = (int)(q - s_pre[ci]);
					const uint64_t hr = s_h0[r];
					const uint32_t wgt = s_w[r];
					const int hp = sk_hdr_prev(hr), nr = sk_hdr_n(hr);
					const int len = hp + nr + K - 1 + sk_hdr_next(hr);
					uint32_t prev, next;
					const Key<NW> key = chop_record<NW>(words, r * BW * 32, len, hp + j, K, prev, next);
					const int s = sk_lds_locate<NW, SLOTS>(s_key, fillc, key, maxfill, max_probe);
					if (s >= 0) {
						sk_lds_update(s_f, s, prev, next, wgt);
						if (TRACK) {
							const uint64_t ord = (sk_hdr_read(hr) << 16) | (uint64_t)(sk_hdr_pos(hr) + (uint32_t)j);
							if (ord < *(volatile unsigned long long *)&s_ord[s])
								atomicMin(&s_ord[s], (unsigned long long)ord);
						}
					} else {
						s_spilled = 1;                   // this item's keys have met memory-side atomics: its merges must be atomics too
						// (never in the round behind an owned flush, see `guard`: every plain store of that flush has landed -- its wave
						// waited for it before the barrier that ended that round)
						const uint64_t ord = TRACK ? ((sk_hdr_read(hr) << 16) | (uint64_t)(sk_hdr_pos(hr) + (uint32_t)j)) : ORD_NONE;
						uint32_t cl = 0;
						atomicAdd(&s_stat[ST_SPILLS], 1u);
						if (!table_merge<NW>(tbl, key, sk_weighted_val(wgt, prev, next), 0u, cl, ord))
							atomicAdd(&s_stat[ST_FAILED], 1u);
						if (cl)
							atomicAdd(&s_stat[ST_CLAIMED], cl);
					}
				}
				if (guard) {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores of the flush have landed ...
					stores_pending = false;
				}
				__syncthreads();                             // ... and behind this barrier everybody's have
				const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)*fillc);     // (nobody adds to it before the round after next)
				const uint32_t fill1 = cur + ko;
				km = ko | (~km & 0x80000000u);               // the next round adds to the other counter
				ko = cur;
				// k-mers per free slot in a round follow the data: a round that used more than half of the free slots halves them (the
				// next one might have run out: its k-mers would take the direct path and cost the item its plain merges), a full
				// round that used less than an eighth doubles them
				if (NW > 1) {                                // (1-word keys, K <= 31: 4 per slot has always been enough, and the bookkeeping costs 5 %)
					if ((fill1 - fill0) * 2u > MAXFILL - fill0)
						room_shift = room_shift ? room_shift - 1u : 0u;
					else if (qe - qb == room && (fill1 - fill0) * 8u < MAXFILL - fill0 && room_shift < 2u)
						room_shift++;
				}
				qb = qe;
				want_flush = (qb >= total && last_tile) || fill1 >= FLUSH_AT;     // the item is done: the table must be clear for the next one
			}
			SK_TICK(2);
		}
	}
#undef SK_TICK
	__syncthreads();
	if (tid == 0) {
		if (s_stat[ST_CLAIMED]) atomicAdd(&stats->distinct, (unsigned long long)s_stat[ST_CLAIMED]);
		if (s_stat[ST_FAILED]) atomicAdd(&stats->probe_fail, (unsigned long long)s_stat[ST_FAILED]);
		if (s_stat[ST_MERGES]) atomicAdd(&stats->sk_merges, (unsigned long long)s_stat[ST_MERGES]);
		if (s_stat[ST_SPILLS]) atomicAdd(&stats->sk_spills, (unsigned long long)s_stat[ST_SPILLS]);
		if (s_stat[ST_GENS]) atomicAdd(&stats->sk_gens, (unsigned long long)s_stat[ST_GENS]);
		if (s_stat[ST_KMERS]) {
			atomicAdd(&stats->kmers, (unsigned long long)s_stat[ST_KMERS]);
			atomicAdd(&stats->sk_counted, (unsigned long long)s_stat[ST_KMERS]);
		}
		if (s_stat[ST_RECS]) atomicAdd(&stats->sk_records, (unsigned long long)s_stat[ST_RECS]);
		if (s_stat[ST_DRECS]) atomicAdd(&stats->sk_distinct_recs, (unsigned long long)s_stat[ST_DRECS]);
		if (s_stat[ST_DKMERS]) atomicAdd(&stats->sk_distinct_kmers, (unsigned long long)s_stat[ST_DKMERS]);
#ifdef SDT_SK_TICKS
		for (int i = 0; i < 4; i++)
			atomicAdd(&stats->sk_cyc[i], cyc[i]);
#endif
	}
}
