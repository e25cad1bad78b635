// sdt_bm_kernels.cuh -- building the bucket-major node table (sdt_table.cuh: BmDir) out of what pass 1 left behind (round 5).
// Included by sdt_gpu.hip after sdt_superkmer_kernels.cuh (LogDesc, sk_lds_locate).
//
// put_kmerset (newhash.c:411-462) is "find the node, add one": update_kmer's saturating link counters (newhash.c:71-96) and
// the count.  The locality pipeline counts a bucket's k-mers in LDS and, since this round, leaves every generation of its LDS
// table as a SEGMENT of (key, val[, ordinal]) entries in the node log (k_sk_count).  A key has one entry per generation it was
// seen in -- 2.5 on the 200 M-read workload -- and ALL of them lie in segments of ONE bucket, because the bucket is a function of
// the key's minimizer.  So the node table is the per-bucket merge of the log:
//
//   k_bm_desc_hist / _place     counting sort of the segment descriptors by bucket (the scan is k_sk_scan)
//   k_bm_flat_hist / _place     nodes that went into the flat table meanwhile (the direct kernel family on small first batches,
//                               records that found no chunk) are sorted by bucket the same way: minimizer per node
//   k_bm_class_hist / _place    buckets in order of falling size (a launch ends when its slowest workgroup does)
//   k_bm_finalize               one workgroup per bucket: every entry of the bucket's segments (+ the bucket's nodes of an earlier
//                               bucket-major table + its flat nodes) into an LDS hash table -- min(63, a + b) per link counter and
//                               the sum of the counts, exactly what replaying the occurrences one by one leaves (sdt_table.cuh:
//                               node_merge) --, then the table is written out ONCE as the bucket's own open-addressing table
//                               (load <= 3/4), slots claimed in an LDS bitmap, empties included.  A bucket with more keys than
//                               the LDS table holds is done in `parts` passes over its input, pass p taking the keys whose hash
//                               falls into part p and writing table p of the bucket (all of one size, fixed after pass 0).
// HBM traffic: the log once (16..48 B per entry, streamed), the table once (streamed).  No memory-side atomic per node.
#pragma once

// a node of the flat table on its way into the merge: key words, val, aux (count bits 31..16, linear, deleted), ordinal
template <int NW> struct BmX { static constexpr int W = NW + 3; };

template <int NW, bool TRACK> struct BmGeo {
	static constexpr int T = NW == 1 ? 1024 : 512;                                     // two workgroups per CU; 1-word keys: 32 waves, what a CU holds (64 registers)
	static constexpr int WAVES_PER_SIMD = 2 * (T / 256);
	static constexpr int MAXD = 128;                                                   // segment descriptors held in LDS at a time
	static constexpr int SLOT_BYTES = NW * 8 + 8 + 4 + (TRACK ? 8 : 0);
	static constexpr int M = (70 * 1024 / SLOT_BYTES) / 64 * 64;                       // LDS merge table: 3584 / 2560 / 2560 / 1984 / 1600 / 1344 slots
	static constexpr uint32_t CAP = (uint32_t)M * 3u / 4u;                             // keys it takes
	static constexpr int BITW = M / 16;                                                // bitmap of a table being written: 2 M slots at most
	static constexpr size_t SMEM = (size_t)M * SLOT_BYTES + (size_t)BITW * 4 + (size_t)MAXD * 8 + (size_t)(MAXD + 1) * 4 + 8;
};

template <int NW> struct BmIn {
	const LogDesc *desc;               // segment descriptors sorted by bucket
	const uint32_t *doff;              // SK_NBF + 1: a bucket's descriptors
	const unsigned long long *dpre;    // SK_NBF + 1: prefix of the entries per bucket
	Table<NW> old;                     // bucket-major table of an earlier finalize (old.dir == nullptr: none)
	const uint32_t *old_cnt;           // SK_NBF: its nodes per bucket
	const uint64_t *xent;              // nodes of the flat table sorted by bucket, BmX<NW>::W words each (nullptr: none)
	const uint32_t *xoff;              // SK_NBF + 1
};
template <int NW> struct BmOut {
	Entry<NW> *ent;
	uint32_t *aux;
	uint64_t *first;
	BmDir *dir;                        // SK_NBF
	uint32_t *cnt;                     // SK_NBF: nodes per bucket
	unsigned long long *ctl;           // [0] slots handed out (may pass `cap`: then nothing was written for the bucket that asked), [1] nodes,
	                                   // [2] buckets done again (LDS table or a part over-full), [3] largest `parts`
	uint64_t cap;                      // slots of ent / aux / first
};
enum { BM_CTL_SLOTS, BM_CTL_NODES, BM_CTL_RESTARTS, BM_CTL_MAXPARTS, BM_CTL_EXT, BM_CTL_UNITS, BM_CTL_N };      // [4] directory entries handed out behind the SK_NBF bucket entries, [5] work units

// A bucket whose input passes BM_GIANT entries is cut, by the top bits of its keys' hashes, into 2^lg SUB-BUCKETS that are
// merged side by side by different workgroups (one workgroup taking the whole of a giant minimizer's bucket -- 84 passes over
// half a million entries on the 200 M-read workload -- was the whole tail of the kernel: 89 ms).  Its directory entry then says
// {base = index of the first of 2^lg directory entries of its own, ssub = 0, parts = lg} (sdt_table.cuh: probe_begin).
constexpr unsigned long long BM_GIANT = 16384;       // entries + nodes a bucket is merged as one unit up to
constexpr unsigned long long BM_SUB_TARGET = 8192;   // ... and what a sub-bucket holds on average beyond
constexpr uint32_t BM_MAX_LG = 12;
struct BmKnobs {                                     // (the host's copy of the three: tests shrink them so that small inputs take every path)
	unsigned long long giant, sub_target;
	uint32_t lds_cap;                                // keys the LDS merge table takes at most (0: what it holds)
};
struct BmUnit { uint32_t b, sub, lg, dirix; };       // bucket, sub-bucket of 2^lg (lg = 0: the whole bucket), its directory entry

__device__ inline unsigned long long bm_bucket_input(const unsigned long long *dpre, const uint32_t *old_cnt, const uint32_t *xoff, uint32_t b)
{
	unsigned long long n = dpre[b + 1] - dpre[b];
	if (old_cnt) n += old_cnt[b];
	if (xoff) n += xoff[b + 1] - xoff[b];
	return n;
}
__device__ inline uint32_t bm_bucket_lg(unsigned long long n, const BmKnobs &kn)
{
	if (n <= kn.giant)
		return 0;
	const unsigned long long want = (n + kn.sub_target - 1) / kn.sub_target;
	if (want < 2)
		return 1;
	uint32_t lg = 64u - (uint32_t)__clzll((long long)(want - 1));                   // ceil(log2(want))
	return lg > BM_MAX_LG ? BM_MAX_LG : lg;
}

__global__ __launch_bounds__(256) void k_bm_desc_hist(const LogDesc *__restrict__ d, const unsigned long long *__restrict__ n_ptr, uint64_t n_cap,
                                                      uint32_t *__restrict__ cnt, unsigned long long *__restrict__ ents)
{
	const uint64_t n = *n_ptr < n_cap ? *n_ptr : n_cap;
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
		const LogDesc x = d[i];
		atomicAdd(&cnt[x.bucket], 1u);
		atomicAdd(&ents[x.bucket], (unsigned long long)x.count);
	}
}

__global__ __launch_bounds__(256) void k_bm_desc_place(const LogDesc *__restrict__ d, const unsigned long long *__restrict__ n_ptr, uint64_t n_cap,
                                                       const uint32_t *__restrict__ off, uint32_t *__restrict__ fill, LogDesc *__restrict__ sorted)
{
	const uint64_t n = *n_ptr < n_cap ? *n_ptr : n_cap;
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
		const LogDesc x = d[i];
		sorted[off[x.bucket] + atomicAdd(&fill[x.bucket], 1u)] = x;
	}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_bm_flat_hist(Table<NW> flat, int K, uint32_t *__restrict__ cnt)
{
	const uint64_t slots = flat.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = flat.ent[s];
		if (e.key[0] == KEY_EMPTY)
			continue;
		Key<NW> key;
#pragma unroll
		for (int i = 0; i < NW; i++)
			key.w[i] = e.key[i];
		atomicAdd(&cnt[key_final_bucket<NW>(key, K)], 1u);
	}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_bm_flat_place(Table<NW> flat, int K, const uint32_t *__restrict__ off, uint32_t *__restrict__ fill, uint64_t *__restrict__ xent)
{
	constexpr int XW = BmX<NW>::W;
	const uint64_t slots = flat.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = flat.ent[s];
		if (e.key[0] == KEY_EMPTY)
			continue;
		Key<NW> key;
#pragma unroll
		for (int i = 0; i < NW; i++)
			key.w[i] = e.key[i];
		const uint32_t b = key_final_bucket<NW>(key, K);
		uint64_t *x = xent + ((size_t)off[b] + atomicAdd(&fill[b], 1u)) * XW;
#pragma unroll
		for (int i = 0; i < NW; i++)
			x[i] = e.key[i];
		x[NW] = e.val;
		x[NW + 1] = flat.aux[s];
		x[NW + 2] = flat.first ? flat.first[s] : ORD_NONE;
	}
}

// size class of a unit = of its bucket's input (every unit of a bucket reads all of it): position of the highest bit, 0 = nothing
__global__ __launch_bounds__(256) void k_bm_class_hist(const unsigned long long *__restrict__ dpre, const uint32_t *__restrict__ old_cnt, const uint32_t *__restrict__ xoff,
                                                       uint32_t nb, uint32_t *__restrict__ ccnt, BmKnobs kn)
{
	__shared__ uint32_t s_c[65];
	if (threadIdx.x < 65) s_c[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nb; b += gridDim.x * 256u) {
		const unsigned long long n = bm_bucket_input(dpre, old_cnt, xoff, b);
		if (n)
			atomicAdd(&s_c[64u - (uint32_t)__clzll((long long)n)], 1u << bm_bucket_lg(n, kn));
	}
	__syncthreads();
	if (threadIdx.x < 65 && s_c[threadIdx.x])
		atomicAdd(&ccnt[threadIdx.x], s_c[threadIdx.x]);
}

// (one thread) start of every class in the unit list, largest class first; cfill = 0; the number of units
__global__ void k_bm_class_scan(const uint32_t *__restrict__ ccnt, uint32_t *__restrict__ cstart, uint32_t *__restrict__ cfill, unsigned long long *__restrict__ ctl)
{
	uint32_t acc = 0;
	for (int c = 64; c >= 0; c--) {
		cstart[c] = acc;
		cfill[c] = 0;
		acc += ccnt[c];
	}
	ctl[BM_CTL_UNITS] = acc;
}

// the unit list; the directory entries of empty and of giant buckets (a giant bucket's sub-buckets get entries of their own behind
// the SK_NBF bucket entries: ext_cap of them, sized by the host from the total input)
__global__ __launch_bounds__(256) void k_bm_class_place(const unsigned long long *__restrict__ dpre, const uint32_t *__restrict__ old_cnt, const uint32_t *__restrict__ xoff,
                                                        uint32_t nb, const uint32_t *__restrict__ cstart, uint32_t *__restrict__ cfill, BmUnit *__restrict__ units,
                                                        BmDir *__restrict__ dir, uint32_t *__restrict__ cnt, unsigned long long *__restrict__ ctl, uint32_t ext_cap, Stats *stats,
                                                        BmKnobs kn)
{
	for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nb; b += gridDim.x * 256u) {
		const unsigned long long n = bm_bucket_input(dpre, old_cnt, xoff, b);
		cnt[b] = 0;
		if (!n) {
			dir[b] = BmDir{0, 0, 0};
			continue;
		}
		const uint32_t c = 64u - (uint32_t)__clzll((long long)n), lg = bm_bucket_lg(n, kn), S = 1u << lg;
		const uint32_t at = cstart[c] + atomicAdd(&cfill[c], S);
		uint32_t ext = 0;
		if (lg) {
			ext = (uint32_t)atomicAdd(&ctl[BM_CTL_EXT], (unsigned long long)S);
			if (ext + S > ext_cap) {                     // (the host's bound: cannot happen)
				atomicAdd(&stats->probe_fail, 1ULL);
				ext = 0;
			}
			dir[b] = BmDir{(uint64_t)nb + ext, 0, lg};
		}
		for (uint32_t j = 0; j < S; j++)
			units[at + j] = BmUnit{b, j, lg, lg ? nb + ext + j : b};
	}
}

// `add` (the node table's val layout: count low 16 | r_links | l_links, fields clamped) and `auxadd` (count bits 31..16, linear,
// deleted) into an LDS node: sdt_table.cuh's node_merge on LDS words
// (the pointers are cast to the LDS address space by hand: left generic, the 4-word instantiation made the backend emit an
// is-this-shared test it could not encode -- "V_CMP_NE_U32 0, src_shared_base: operand has incorrect register class")
typedef __attribute__((address_space(3))) unsigned long long bm_lds_u64;
typedef __attribute__((address_space(3))) uint32_t bm_lds_u32;
__device__ __forceinline__ void bm_lds_merge(unsigned long long *val_g, uint32_t *hi_g, uint64_t add, uint32_t auxadd)
{
	bm_lds_u64 *val = (bm_lds_u64 *)val_g;
	bm_lds_u32 *hi = (bm_lds_u32 *)hi_g;
	unsigned long long seen = __hip_atomic_load(val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	uint32_t c;
	for (;;) {
		// min(63, a + b) for the eight 6-bit fields at once: the low five bits of every field add without leaving it; a field
		// saturates when both top bits are set, or one of them and the carry out of the low five
		constexpr uint64_t H = 0x820820820820ULL, L = 0x7DF7DF7DF7DFULL;
		const uint64_t t = (seen & L) + (add & L), xh = seen & H, yh = add & H;
		const uint64_t sat = ((xh & yh) | ((xh | yh) & t)) & H;
		uint64_t nv = (t | xh | yh | ((sat >> 5) * 63ULL)) & 0xFFFFFFFFFFFFULL;
		c = (uint32_t)(seen >> 48) + (uint32_t)(add >> 48);
		nv |= (uint64_t)(c & 0xFFFFu) << 48;
		if (__hip_atomic_compare_exchange_strong(val, &seen, (unsigned long long)nv, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
			break;                                       // (a failed exchange leaves the current value in `seen`)
	}
	const uint32_t up = (c >> 16) + (auxadd & 0xFFFFu);
	if (up)
		(void)__hip_atomic_fetch_add(hi, up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	if (auxadd & (AUX_LINEAR | AUX_DELETED))
		(void)__hip_atomic_fetch_or(hi, auxadd & (AUX_LINEAR | AUX_DELETED), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one entry into the LDS merge table: a unit takes the keys of its sub-bucket (the top lg bits of the hash's high half), pass p of
// P those whose following bits fall into part p; no room: *abort = 1
template <int NW, bool TRACK, int M>
__device__ __forceinline__ void bm_insert(unsigned long long *m_key, unsigned long long *m_val, unsigned long long *m_ord, uint32_t *m_hi, uint32_t *fill, uint32_t *abort_flag,
                                          uint32_t cap, uint32_t lg, uint32_t sub, uint32_t P, uint32_t p, const Key<NW> &key, uint64_t val, uint32_t auxadd, uint64_t ord)
{
	if (lg || P > 1) {
		uint32_t hh = bm_hash_hi(bm_fold<NW>(key));
		if (lg) {
			if ((hh >> (32u - lg)) != sub)
				return;
			hh <<= lg;
		}
		if (P > 1 && __umulhi(hh, P) != p)
			return;
	}
	const int s = sk_lds_locate<NW, M>(m_key, fill, key, cap);
	if (s < 0) {
		*abort_flag = 1;
		return;
	}
	bm_lds_merge(&m_val[s], &m_hi[s], val, auxadd);
	if (TRACK && ord != ORD_NONE)
		atomicMin(&m_ord[s], (unsigned long long)ord);
}

constexpr unsigned long long BM_CHUNK = 65536;          // slots a workgroup takes from the table at a time (a global atomic per unit was a round trip on every unit's critical path)

template <int NW> __device__ __forceinline__ void bm_store_empty(const BmOut<NW> &out, uint64_t slot, bool track)
{
	Entry<NW> e;
#pragma unroll
	for (int wv = 0; wv < NW; wv++)
		e.key[wv] = KEY_EMPTY;
	e.val = 0;
	if constexpr (NW != 1) e.pad = 0;
	out.ent[slot] = e;
	out.aux[slot] = 0;
	if (track) out.first[slot] = ORD_NONE;
}

// Units are dealt out round robin (the list is in order of falling size: that IS a fair deal), so everything a workgroup needs to
// know about its next unit sits at addresses the scalar unit can compute ahead of time -- no atomic, no broadcast, no dependent
// vector load between two units.  Slots come out of a chunk of the table the workgroup owns (BM_CHUNK at a time; what is left of
// a chunk when it is given up is filled with empties: the scans of the table must find nothing there).
template <int NW, bool TRACK>
__global__ __launch_bounds__((BmGeo<NW, TRACK>::T), (BmGeo<NW, TRACK>::WAVES_PER_SIMD)) void k_bm_finalize(BmIn<NW> in, BmOut<NW> out, const BmUnit *__restrict__ units,
                                                                                                         int K, Stats *stats, BmKnobs kn)
{
	using G = BmGeo<NW, TRACK>;
	constexpr int M = G::M, T = G::T, BITW = G::BITW, MAXD = G::MAXD;
	const uint32_t CAP = kn.lds_cap && kn.lds_cap < G::CAP ? kn.lds_cap : G::CAP;
	constexpr int LW = NW + 1 + (TRACK ? 1 : 0), XW = BmX<NW>::W;
	extern __shared__ unsigned long long bm_sm[];
	unsigned long long *m_key = bm_sm;                               // NW x M, word-major (sk_lds_locate's layout)
	unsigned long long *m_val = m_key + NW * M;                      // M
	unsigned long long *m_ord = m_val + M;                           // M when TRACK
	uint32_t *m_hi = (uint32_t *)(m_ord + (TRACK ? M : 0));          // M: the node's aux word
	uint32_t *s_bits = m_hi + M;                                     // BITW
	unsigned long long *s_dptr = (unsigned long long *)(s_bits + BITW + ((BITW + M) & 1));      // MAXD: the segments being read (8-byte aligned)
	uint32_t *s_dpre = (uint32_t *)(s_dptr + MAXD);                  // MAXD + 1: prefix of their entry counts
	__shared__ uint32_t s_fill, s_abort;
	__shared__ unsigned long long s_base;
	const int tid = threadIdx.x;
	const uint32_t nunits = (uint32_t)out.ctl[BM_CTL_UNITS];
	uint32_t restarts = 0, maxparts = 0;
	unsigned long long nodes = 0;
#ifdef SDT_BM_TICKS
	unsigned long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = wall_clock64(), t1;
#define BM_TICK(i) do { t1 = wall_clock64(); cyc[i] += t1 - t0; t0 = t1; } while (0)
#else
#define BM_TICK(i) do { } while (0)
#endif
	unsigned long long loc_next = 0, loc_end = 0;                    // (uniform) the workgroup's chunk of the table: [loc_next, loc_end)
	// the merge table and the bitmap start out clear (and are cleared again behind every part: see the end of the part loop)
	for (int i = tid; i < M; i += T) {
		m_key[i] = KEY_EMPTY;
		m_val[i] = 0;
		m_hi[i] = 0;
		if (TRACK) m_ord[i] = ORD_NONE;
	}
	for (int i = tid; i < BITW; i += T)
		s_bits[i] = 0;
	if (tid == 0) {
		s_fill = 0;
		s_abort = 0;
	}
	__syncthreads();
	// slots for a unit: out of the chunk; a new chunk (or, for a large unit, exactly what it needs) by one global atomic.  A chunk
	// past the end of the table is handed out all the same, but nothing is written there (`valid`): BM_CTL_SLOTS then ends up at
	// exactly what a run with enough room takes -- the host runs the kernel again with that.
	bool loc_valid = false;                              // (uniform)
	auto take = [&](unsigned long long need, bool &valid) -> unsigned long long {
		if (loc_next + need <= loc_end) {
			const unsigned long long at = loc_next;
			loc_next += need;
			valid = loc_valid;
			return at;
		}
		const bool big = need > BM_CHUNK / 4;
		if (!big && loc_valid) {
			// give the rest of the old chunk up: empties
			for (unsigned long long sl = loc_next + (unsigned long long)tid; sl < loc_end; sl += T)
				bm_store_empty<NW>(out, sl, TRACK);
		}
		__syncthreads();                                 // (s_base may still be read from the last time)
		if (tid == 0)
			s_base = atomicAdd(&out.ctl[BM_CTL_SLOTS], big ? need : BM_CHUNK);
		__syncthreads();
		const unsigned long long got = s_base;
		valid = got + (big ? need : BM_CHUNK) <= out.cap;
		if (!big) {
			loc_next = got + need;
			loc_end = got + BM_CHUNK;
			loc_valid = valid;
		}
		return got;
	};
	for (uint32_t at = blockIdx.x; at < nunits; at += gridDim.x) {
		const BmUnit u = units[at];
		const uint32_t b = u.b, lg = u.lg, sub = u.sub;
		const uint32_t d0 = in.doff[b], d1 = in.doff[b + 1];
		const unsigned long long nlog = in.dpre[b + 1] - in.dpre[b];
		// the bucket's nodes in the table of an earlier finalize: one range, or one per sub-bucket it was cut into then
		BmDir od = {0, 0, 0};
		uint32_t nold = 0, nold_r = 0;
		if (in.old.dir) {
			od = in.old.dir[b];
			nold = in.old_cnt[b];
			nold_r = od.parts ? (od.ssub ? 1u : 1u << od.parts) : 0u;
		}
		const uint32_t x0 = in.xent ? in.xoff[b] : 0u, x1 = in.xent ? in.xoff[b + 1] : 0u;
		// (what this unit takes of the bucket's input: all of it, or about a 2^lg-th)
		const unsigned long long n_in = (nlog + nold + (x1 - x0)) >> lg;
		// parts: one when everything fits whatever the keys are; else from a guess of the distinct keys (an entry of the log is one of
		// ~2.5 of its key, a node of a table is the only one), doubled whenever the LDS table fills up all the same
		uint32_t P = 1;
		if (n_in > CAP || lg) {
			const unsigned long long est = ((nold + (x1 - x0) + nlog * 9 / 20) >> lg) + 1;       // (an entry of the log is one of ~2.5 of its key)
			const unsigned long long per = (unsigned long long)CAP * 90 / 100;
			P = (uint32_t)((est + per - 1) / per);
		}
		uint32_t margin = 0, floor_ssub = 8;
		BM_TICK(0);                                      // unit header
		unsigned long long base = 0;
		uint32_t ssub = 0, total_d = 0;
		bool valid = false;                          // the unit's slots lie inside the table
		for (;;) {
			bool ok = true, lds_full = false;
			base = 0;
			ssub = 0;
			total_d = 0;
			valid = false;
			bool allocated = false;
			for (uint32_t p = 0; p < P; p++) {
				// ---- the bucket's segments of the log: descriptors into LDS (MAXD at a time), then ONE loop over all their entries
				// with two loads per lane in flight (a loop per segment paid a descriptor round trip and a ragged last sweep per
				// segment: 46 of the 70 us a bucket took)
				for (uint32_t dc = d0; dc < d1; dc += MAXD) {
					const uint32_t nd = d1 - dc < (uint32_t)MAXD ? d1 - dc : (uint32_t)MAXD;
					if (dc != d0)
						__syncthreads();                 // (the sweeps over the last lot are over)
					if ((uint32_t)tid < nd) {
						const LogDesc ds = in.desc[dc + tid];
						s_dptr[tid] = ds.ptr;
						s_dpre[tid + 1] = ds.count;
					}
					if (tid == 0)
						s_dpre[0] = 0;
					__syncthreads();
					if (tid < 64) {                      // inclusive scan of <= 128 counts by one wave (two per lane)
						const uint32_t a = 2u * tid < nd ? s_dpre[2 * tid + 1] : 0u, c2 = 2u * tid + 1u < nd ? s_dpre[2 * tid + 2] : 0u;
						uint32_t x = a + c2;
#pragma unroll
						for (int dd = 1; dd < 64; dd <<= 1) {
							const uint32_t y = __shfl_up(x, dd);
							if (tid >= dd)
								x += y;
						}
						if (2u * tid < nd) s_dpre[2 * tid + 1] = x - c2;
						if (2u * tid + 1u < nd) s_dpre[2 * tid + 2] = x;
					}
					__syncthreads();
					BM_TICK(1);                          // descriptors + scan
					const uint32_t total = s_dpre[nd];
					uint32_t seg = 0;
					for (uint32_t e0 = 0; e0 < total; e0 += 2u * T) {
						const uint32_t ea = e0 + (uint32_t)tid, eb = ea + (uint32_t)T;
						Key<NW> ka, kb;
						uint64_t va = 0, vb = 0, oa = ORD_NONE, ob = ORD_NONE;
						const bool ha = ea < total, hb = eb < total;
						if (ha) {
							while (ea >= s_dpre[seg + 1]) seg++;
							const uint64_t *x = (const uint64_t *)s_dptr[seg] + (size_t)(ea - s_dpre[seg]) * LW;
							if (NW == 1 && !TRACK) {
								const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(x);
								ka.w[0] = kv.x;
								va = kv.y;
							} else {
#pragma unroll
								for (int i = 0; i < NW; i++)
									ka.w[i] = x[i];
								va = x[NW];
								if (TRACK) oa = x[NW + (TRACK ? 1 : 0)];
							}
						}
						if (hb) {
							while (eb >= s_dpre[seg + 1]) seg++;
							const uint64_t *x = (const uint64_t *)s_dptr[seg] + (size_t)(eb - s_dpre[seg]) * LW;
							if (NW == 1 && !TRACK) {
								const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(x);
								kb.w[0] = kv.x;
								vb = kv.y;
							} else {
#pragma unroll
								for (int i = 0; i < NW; i++)
									kb.w[i] = x[i];
								vb = x[NW];
								if (TRACK) ob = x[NW + (TRACK ? 1 : 0)];
							}
						}
						if (ha) bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_fill, &s_abort, CAP, lg, sub, P, p, ka, va, 0u, oa);
						if (hb) bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_fill, &s_abort, CAP, lg, sub, P, p, kb, vb, 0u, ob);
					}
				}
				BM_TICK(2);                              // entries of the log
				// ---- its nodes in the table of an earlier finalize (one range, or one per sub-bucket it was cut into then), and its
				// nodes out of the flat table
				for (uint32_t src = 0; src < nold_r + 1u; src++) {
					const int kind = src < nold_r ? 1 : 2;
					uint64_t cnt = 0, obase = 0;
					if (kind == 1) {
						const BmDir r = od.ssub ? od : in.old.dir[od.base + src];
						obase = r.base;
						cnt = r.parts ? (uint64_t)r.parts * r.ssub : 0;
					} else {
						cnt = x1 - x0;
					}
					for (uint64_t e = (uint64_t)tid; e < cnt; e += T) {
						Key<NW> key;
						uint64_t val, ord = ORD_NONE;
						uint32_t auxadd = 0;
						if (kind == 1) {
							const uint64_t sl = obase + e;
							const Entry<NW> *oe = in.old.ent + sl;
							if (oe->key[0] == KEY_EMPTY)
								continue;
#pragma unroll
							for (int i = 0; i < NW; i++)
								key.w[i] = oe->key[i];
							val = oe->val;
							auxadd = in.old.aux[sl];
							if (TRACK) ord = in.old.first[sl];
						} else {
							const uint64_t *x = in.xent + (size_t)(x0 + e) * XW;
#pragma unroll
							for (int i = 0; i < NW; i++)
								key.w[i] = x[i];
							val = x[NW];
							auxadd = (uint32_t)x[NW + 1];
							ord = x[NW + 2];
						}
						bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_fill, &s_abort, CAP, lg, sub, P, p, key, val, auxadd, ord);
					}
				}
				__syncthreads();
				BM_TICK(3);                              // old + flat nodes, the barrier behind the inserts
				const uint32_t d = s_fill;
				lds_full = s_abort != 0;
				bool part_ok = !lds_full;
				if (part_ok && p == 0) {
					// slots per table: load 3/4; with several parts a quarter (and more after a failed attempt) on top of what part 0 holds
					uint64_t want = P == 1 ? ((uint64_t)d * 4 + 2) / 3 + 1 : ((uint64_t)d * (5 + margin) / 4 * 4 + 2) / 3 + 16;
					want = (want + 7) & ~7ULL;
					if (want < floor_ssub) want = floor_ssub;
					if (want > (uint64_t)BITW * 32) want = (uint64_t)BITW * 32;
					ssub = (uint32_t)want;
					base = take((unsigned long long)P * ssub, valid);       // (uniform: every lane keeps the chunk's cursor)
					allocated = true;
				}
				if (part_ok && (uint64_t)d * 16 > (uint64_t)ssub * 15) {     // (one empty slot at the very least: a look-up of an absent key must end)
					part_ok = false;
					floor_ssub = (uint32_t)((((uint64_t)d * 5 / 3 + 16) + 7) & ~7ULL);      // what this part needs, and a quarter
				}
				BM_TICK(4);                              // slots
				if (part_ok) {
					total_d += d;
					if (valid) {
						// write the table of this part: every node claims the first free slot from its home on, in the bitmap
						const uint64_t sub0 = base + (uint64_t)p * ssub;
						for (int i = tid; i < M; i += T) {
							if (m_key[i] == KEY_EMPTY)
								continue;
							Key<NW> key;
							key.w[0] = m_key[i];
#pragma unroll
							for (int wv = 1; wv < NW; wv++)
								key.w[wv] = m_key[wv * M + i];
							uint32_t pos = __umulhi(bm_hash_lo(bm_fold<NW>(key)), ssub);
							for (;;) {
								const uint32_t bit = 1u << (pos & 31u);
								if (!(atomicOr(&s_bits[pos >> 5], bit) & bit))
									break;
								pos = pos + 1u == ssub ? 0u : pos + 1u;
							}
							Entry<NW> e;
#pragma unroll
							for (int wv = 0; wv < NW; wv++)
								e.key[wv] = key.w[wv];
							e.val = m_val[i];
							if constexpr (NW != 1) e.pad = 0;
							out.ent[sub0 + pos] = e;
							out.aux[sub0 + pos] = m_hi[i];
							if (TRACK) out.first[sub0 + pos] = m_ord[i];
						}
						__syncthreads();
						BM_TICK(5);                      // nodes written
						for (uint32_t pos = (uint32_t)tid; pos < ssub; pos += T)
							if (!(s_bits[pos >> 5] & (1u << (pos & 31u))))
								bm_store_empty<NW>(out, sub0 + pos, TRACK);
					}
				}
				// the merge table and the bitmap are clear again for whatever comes next (the next part, the next attempt, the next unit)
				__syncthreads();
				for (int i = tid; i < M; i += T) {
					m_key[i] = KEY_EMPTY;
					m_val[i] = 0;
					m_hi[i] = 0;
					if (TRACK) m_ord[i] = ORD_NONE;
				}
				for (int i = tid; i < BITW; i += T)
					s_bits[i] = 0;
				if (tid == 0) {
					s_fill = 0;
					s_abort = 0;
				}
				__syncthreads();
				BM_TICK(6);                              // empties, LDS cleared
				if (!part_ok) {
					ok = false;
					break;
				}
			}
			if (ok)
				break;
			// again, with more parts (the LDS table filled up) or more room per part (a part turned out larger than part 0 suggested).
			// Slots handed out to the failed attempt are wiped: the scans of the table must find nothing in them.
			if (allocated && valid) {
				const uint64_t hi = (uint64_t)P * ssub;
				for (uint64_t pos = (uint64_t)tid; pos < hi; pos += T)
					bm_store_empty<NW>(out, base + pos, TRACK);
			}
			if (lds_full) P *= 2; else margin += 2;
			restarts++;
		}
		if (tid == 0) {
			out.dir[u.dirix] = valid && total_d ? BmDir{base, ssub, P} : BmDir{0, 0, 0};
			out.cnt[u.dirix] = total_d;
			if (lg && total_d)
				atomicAdd(&out.cnt[b], total_d);     // (the bucket's own entry counts all its sub-buckets: k_bm_class_place zeroed it)
		}
		nodes += total_d;
		maxparts = P > maxparts ? P : maxparts;
	}
	// what is left of the last chunk
	if (loc_valid)
		for (unsigned long long sl = loc_next + (unsigned long long)tid; sl < loc_end; sl += T)
			bm_store_empty<NW>(out, sl, TRACK);
	if (tid == 0) {
		if (nodes) atomicAdd(&out.ctl[BM_CTL_NODES], nodes);
		if (restarts) atomicAdd(&out.ctl[BM_CTL_RESTARTS], (unsigned long long)restarts);
		atomicMax(&out.ctl[BM_CTL_MAXPARTS], (unsigned long long)maxparts);
#ifdef SDT_BM_TICKS
		for (int i = 0; i < 4; i++) {
			atomicAdd(&stats->sk_cyc[i], cyc[i]);
			atomicAdd(&stats->sk_cyc1[i], cyc[4 + i]);
		}
#endif
	}
#undef BM_TICK
}
