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
#ifndef BM_WGS_PER_CU
#define BM_WGS_PER_CU 1
#endif
#ifndef BM_T1
#define BM_T1 1024
#endif
	static constexpr int T = BM_WGS_PER_CU == 1 ? 1024 : (NW == 1 ? BM_T1 : 512);             // two workgroups per CU; 1-word keys: 32 waves, what a CU holds (64 registers)
	static constexpr int WAVES_PER_SIMD = BM_WGS_PER_CU * (T / 256);
	static constexpr int MAXD = 128;                                                   // segment descriptors held in LDS at a time
	static constexpr int SLOT_BYTES = NW * 8 + 8 + 4 + (TRACK ? 8 : 0);
	static constexpr int M = ((BM_WGS_PER_CU == 1 ? 144 : 70) * 1024 / SLOT_BYTES) / 512 * 512;                     // LDS image of a table: 3584 / 2560 / 2560 / 1536 / 1536 / 1024 slots
	static constexpr int STEP = M / 16;                                                // a bucket's table: a multiple of this many slots (or M / 64 for the smallest)
	static constexpr size_t SMEM = (size_t)M * SLOT_BYTES + (size_t)MAXD * 8 + (size_t)(MAXD + 1) * 4 + 16;
	static_assert(M % 512 == 0, "table sizes are multiples of 8 slots down to M / 64");
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
constexpr unsigned long long BM_GIANT = 32768;       // entries + nodes a bucket is merged as one unit up to (its parts re-read its input)
constexpr unsigned long long BM_SUB_TARGET = 12288;  // ... and what a sub-bucket holds on average beyond (every sub-bucket reads all of it)
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

// Find-or-claim in the LDS image of a bucket's table: `msz` slots (word-major key words, stride M), linear probing from the
// key's home -- the very layout the table has in HBM afterwards (sdt_table.cuh: bm_home), so that the image is written out as it
// stands.  1-word keys probe a group of four slots per step (home is a multiple of four): 32 bytes of keys per LDS round trip;
// a slot that looked empty in the snapshot is settled by the compare-and-swap, and a key never changes once it is there.
// Nothing counts the keys here (one counter for all lanes was a same-address atomic per new key: the LDS serialises those); an
// image that is too full shows as a probe sequence of more than BM_MAX_PROBE slots (-1), and its keys are counted when it is
// written out.
constexpr uint32_t BM_MAX_PROBE = 128;
template <int NW, int M>
__device__ __forceinline__ int bm_locate(unsigned long long *m_key, uint32_t msz, const Key<NW> &key, uint32_t hlo)
{
	uint32_t s = bm_home<NW>(hlo, msz);
	const uint32_t lim = msz < BM_MAX_PROBE ? msz : BM_MAX_PROBE;
	if constexpr (NW == 1) {
		// (per group ONE read of the four keys and at most one compare-and-swap -- at the first slot that looks empty; a key that
		// is in the table sits before the first empty slot of its probe sequence, nothing is ever removed.  Written slot by slot,
		// the lanes of a wave took the four slots' branches one after the other: four times the LDS instructions.)
		for (uint32_t probe = 0; probe < lim; probe += 4) {
			const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&m_key[s]), b = *reinterpret_cast<const ulonglong2 *>(&m_key[s + 2]);
			uint64_t k0 = a.x, k1 = a.y, k2 = b.x, k3 = b.y;
			for (;;) {
				const uint64_t me = key.w[0];
				// position of the key / of the first empty slot in the group (4: none)
				const int jm = k0 == me ? 0 : (k1 == me ? 1 : (k2 == me ? 2 : (k3 == me ? 3 : 4)));
				const int je = k0 == KEY_EMPTY ? 0 : (k1 == KEY_EMPTY ? 1 : (k2 == KEY_EMPTY ? 2 : (k3 == KEY_EMPTY ? 3 : 4)));
				if (jm < je)
					return (int)s + jm;
				if (je == 4)
					break;                               // a full group of other keys: on to the next
				const uint64_t old = atomicCAS(&m_key[s + je], (unsigned long long)KEY_EMPTY, (unsigned long long)me);
				if (old == KEY_EMPTY || old == me)
					return (int)s + je;
				// somebody else's key got there first: it is part of the picture now
				k0 = je == 0 ? old : k0;
				k1 = je == 1 ? old : k1;
				k2 = je == 2 ? old : k2;
				k3 = je == 3 ? old : k3;
			}
			s = s + 4 == msz ? 0u : s + 4;
		}
		return -1;
	} else {
		// Multi-word keys, the same four slots per step: the first key words of a group in one read; a candidate (first word equal)
		// is settled by its other words, an empty slot by a compare-and-swap to KEY_LOCKED, the other words, and the first word last
		// (release).  A slot that is LOCKED is somebody writing those: the group is read again.  Flag form -- the claimer's stores
		// are INSIDE the loop body and the loop ends on `found`: an exit path may be moved behind the loop, and the lanes that wait
		// for the claimer -- possibly a lane of their own wave -- would wait for ever (tools/lds_cursor_stress.hip).
		int found = -2;
		for (uint32_t probe = 0; probe < lim && found == -2;) {
			asm volatile("" ::: "memory");               // (the group is READ AGAIN when a slot was being written: not a value the compiler may keep)
			const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(&m_key[s]), b = *reinterpret_cast<const ulonglong2 *>(&m_key[s + 2]);
			uint64_t k0 = a.x, k1 = a.y, k2 = b.x, k3 = b.y;
			const uint64_t me = key.w[0];
			bool again = false;
			while (found == -2 && !again) {
				// the first slot of the group that matters: a candidate, an empty one, or one being written
				const int j = (k0 == me || k0 >= KEY_LOCKED) ? 0 : ((k1 == me || k1 >= KEY_LOCKED) ? 1 : ((k2 == me || k2 >= KEY_LOCKED) ? 2 : ((k3 == me || k3 >= KEY_LOCKED) ? 3 : 4)));
				if (j == 4)
					break;                               // four other keys: on to the next group
				const uint64_t kj = j == 0 ? k0 : (j == 1 ? k1 : (j == 2 ? k2 : k3));
				uint64_t seen = kj;
				if (kj == KEY_EMPTY) {
					seen = atomicCAS(&m_key[s + j], (unsigned long long)KEY_EMPTY, (unsigned long long)KEY_LOCKED);
					if (seen == KEY_EMPTY) {
#pragma unroll
						for (int i = 1; i < NW; i++)
							m_key[i * M + s + j] = key.w[i];
						__hip_atomic_store(&m_key[s + j], me, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						found = (int)s + j;
						continue;
					}
				}
				if (seen == KEY_LOCKED) {
					again = true;                        // (its claimer is about to publish it: look at the group again)
					continue;
				}
				bool same = seen == me;
				if (same) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
					for (int i = 1; i < NW; i++)
						same = same && (__hip_atomic_load(&m_key[i * M + s + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == key.w[i]);
				}
				if (same) {
					found = (int)s + j;
					continue;
				}
				// another key (its first word may even equal ours): not a candidate any more
				const uint64_t other = 1ULL << 62;       // (nobody's first word: a key word never has its top two bits set; < KEY_LOCKED)
				if (seen == me) {
					k0 = j == 0 ? other : k0; k1 = j == 1 ? other : k1; k2 = j == 2 ? other : k2; k3 = j == 3 ? other : k3;
				} else {
					k0 = j == 0 ? seen : k0; k1 = j == 1 ? seen : k1; k2 = j == 2 ? seen : k2; k3 = j == 3 ? seen : k3;
				}
			}
			if (found == -2 && !again) {
				s = s + 4 == msz ? 0u : s + 4;
				probe += 4;
			}
		}
		return found == -2 ? -1 : found;
	}
}

// one entry into the LDS image: a unit takes the keys of its sub-bucket (the top lg bits of the `hi` hash), pass p of P those whose
// following bits fall into part p; no room: *abort = 1
template <int NW, bool TRACK, int M>
__device__ __forceinline__ void bm_insert(unsigned long long *m_key, unsigned long long *m_val, unsigned long long *m_ord, uint32_t *m_hi, uint32_t *abort_flag,
                                          uint32_t msz, uint32_t lg, uint32_t sub, uint32_t P, uint32_t p, const Key<NW> &key, uint64_t val, uint32_t auxadd, uint64_t ord)
{
	const uint32_t f = bm_fold<NW>(key);
	if (lg || P > 1) {
		uint32_t hh = bm_hash_hi(f);
		if (lg) {
			if ((hh >> (32u - lg)) != sub)
				return;
			hh <<= lg;
		}
		if (P > 1 && __umulhi(hh, P) != p)
			return;
	}
#if defined(SDT_BM_EXP) && (SDT_BM_EXP == 1 || SDT_BM_EXP == 3)                  /* measurement builds: what the phases of an insert cost (profiles/r5) */
	asm volatile("" :: "v"(f), "v"(val));
	return;
#endif
	const int s = bm_locate<NW, M>(m_key, msz, key, bm_hash_lo(f));
	if (s < 0) {
		*abort_flag = 1;
		return;
	}
#if defined(SDT_BM_EXP) && SDT_BM_EXP == 2
	asm volatile("" :: "v"(s), "v"(val));
	return;
#endif
	bm_lds_merge(&m_val[s], &m_hi[s], val, auxadd);
	if (TRACK && ord != ORD_NONE)
		atomicMin(&m_ord[s], (unsigned long long)ord);
}

#ifndef BM_UNROLL
#define BM_UNROLL 2
#endif
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
//
// A unit's table(s) are built IN LDS in the layout they have in HBM -- `msz` slots, linear probing from bm_home -- and written out
// as they stand, empties included: no second hash, no slot claims, no pass over the empties.  `msz` comes from a guess of the
// unit's distinct keys (half its log entries + its nodes): a multiple of M / 16 aimed at a load of 0.7; a unit that needs more
// than M slots gets `parts` tables of M; a guess that turns out short (the image fills past 4/5) costs the unit another attempt
// with twice the room.
template <int NW, bool TRACK>
__global__ __launch_bounds__((BmGeo<NW, TRACK>::T), (BmGeo<NW, TRACK>::WAVES_PER_SIMD)) void k_bm_finalize(BmIn<NW> in, BmOut<NW> out, const BmUnit *__restrict__ units,
                                                                                                         int K, Stats *stats, BmKnobs kn)
{
	using G = BmGeo<NW, TRACK>;
	constexpr int M = G::M, T = G::T, MAXD = G::MAXD;
	constexpr int LW = NW + 1 + (TRACK ? 1 : 0), XW = BmX<NW>::W;
	const uint32_t MEFF = kn.lds_cap && kn.lds_cap < (uint32_t)M ? ((kn.lds_cap + 7u) & ~7u) : (uint32_t)M;     // (tests: a small image forces several parts)
	extern __shared__ __attribute__((aligned(16))) unsigned long long bm_sm[];
	unsigned long long *m_key = bm_sm;                               // NW x M, word-major (16-byte aligned: bm_locate reads two keys at a time)
	unsigned long long *m_val = m_key + NW * M;                      // M
	unsigned long long *m_ord = m_val + M;                           // M when TRACK
	uint32_t *m_hi = (uint32_t *)(m_ord + (TRACK ? M : 0));          // M: the node's aux word
	unsigned long long *s_dptr = (unsigned long long *)(m_hi + M + (M & 1));      // MAXD: the segments being read (8-byte aligned)
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
	// the image starts out clear (and is cleared again as it is written out)
	for (int i = tid; i < M; i += T) {
		m_key[i] = KEY_EMPTY;
		m_val[i] = 0;
		m_hi[i] = 0;
		if (TRACK) m_ord[i] = ORD_NONE;
	}
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
		// Slots.  An attempt that runs out of room costs the unit another pass over its input -- as much as ten thousand slots cost
		// every later scan of the table -- so: a unit whose input fits one image whatever its keys are gets room for all of it (load
		// <= 0.8 even if no two entries share a key); a larger one room for 0.65 distinct keys per log entry (0.4 is the average of deep
		// transcriptome data) at a load of 0.8, in as many tables as that takes, each as small as that allows.
		const unsigned long long n_unit = ((nlog + nold + (x1 - x0)) >> lg) + 1;
		const unsigned long long est = ((nold + (x1 - x0) + nlog * 13 / 20) >> lg) + 1;
		const uint32_t step = (uint32_t)G::STEP < MEFF ? (uint32_t)G::STEP : MEFF;
		uint32_t P = 1, msz = MEFF;
		if (!lg && n_unit * 5 / 4 + 1 <= MEFF) {
			const unsigned long long want = n_unit * 5 / 4 + 1;
			msz = (uint32_t)((want + step - 1) / step) * step;
			if (want <= (uint32_t)M / 64 && (uint32_t)M / 64 <= MEFF) msz = (uint32_t)M / 64;
		} else {
			const unsigned long long want = est * 5 / 4 + 1;
			P = (uint32_t)((want + MEFF - 1) / MEFF);
			msz = (uint32_t)(((want + P - 1) / P + step - 1) / step) * step;
		}
		if (msz > MEFF) msz = MEFF;
		BM_TICK(0);                                      // unit header
		unsigned long long base = 0;
		uint32_t total_d = 0;
		bool valid = false;                              // the unit's slots lie inside the table
		for (;;) {
			bool ok = true;
			total_d = 0;
			base = take((unsigned long long)P * msz, valid);         // (uniform: every lane keeps the chunk's cursor)
			BM_TICK(4);                                  // slots
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
					// (BM_UNROLL entries per lane in flight: the loop is a chain of memory round trips -- address out of LDS, load, insert --
					// and with one or two loads per wave between waits it ran at a fifth of what the memory system gives)
					constexpr int U = BM_UNROLL;
					typedef const __attribute__((address_space(1))) uint64_t *gptr;      // (a pointer out of LDS is generic: say that it is global memory)
					// (software pipeline: the loads of the next U entries are in flight while these U go into the image)
					Key<NW> kk[U], kn[U];
					uint64_t vv[U], oo[U], vn[U], on[U];
					bool hh[U], hn[U];
					auto fetch = [&](uint32_t e0, Key<NW> *k, uint64_t *v, uint64_t *o, bool *h) {
#pragma unroll
						for (int q = 0; q < U; q++) {
							const uint32_t e = e0 + (uint32_t)q * T + (uint32_t)tid;
							h[q] = e < total;
							v[q] = 0;
							o[q] = ORD_NONE;
							if (h[q]) {
								while (e >= s_dpre[seg + 1]) seg++;
#if defined(SDT_BM_EXP) && SDT_BM_EXP == 3                  /* measurement build: every load hits the same few lines */
								gptr x = (gptr)(s_dptr[0]) + (size_t)((e - s_dpre[seg]) & 63u) * LW;
#else
								gptr x = (gptr)(s_dptr[seg]) + (size_t)(e - s_dpre[seg]) * LW;
#endif
								if (NW == 1 && !TRACK) {
									typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
									const u64x2 kv = *reinterpret_cast<const __attribute__((address_space(1))) u64x2 *>(x);
									k[q].w[0] = kv.x;
									v[q] = kv.y;
								} else {
#pragma unroll
									for (int i = 0; i < NW; i++)
										k[q].w[i] = x[i];
									v[q] = x[NW];
									if (TRACK) o[q] = x[NW + (TRACK ? 1 : 0)];
								}
							}
						}
					};
					fetch(0, kk, vv, oo, hh);
					for (uint32_t e0 = 0; e0 < total; e0 += (uint32_t)U * T) {
						const bool more = e0 + (uint32_t)U * T < total;
						if (more)
							fetch(e0 + (uint32_t)U * T, kn, vn, on, hn);
#pragma unroll
						for (int q = 0; q < U; q++)
							if (hh[q]) bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_abort, msz, lg, sub, P, p, kk[q], vv[q], 0u, oo[q]);
#pragma unroll
						for (int q = 0; q < U; q++) {
							kk[q] = kn[q]; vv[q] = vn[q]; oo[q] = on[q]; hh[q] = more && hn[q];
						}
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
						bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_abort, msz, lg, sub, P, p, key, val, auxadd, ord);
					}
				}
				__syncthreads();
				BM_TICK(3);                              // old + flat nodes, the barrier behind the inserts
				const bool part_ok = s_abort == 0;
				__syncthreads();                         // (everybody has read it before it is reset below)
				// the image goes out as it stands (a failed part: only cleared) and is clear again for whatever comes next
				const uint64_t sub0 = base + (uint64_t)p * msz;
				uint32_t mine = 0;
				for (uint32_t i = (uint32_t)tid; i < msz; i += T) {
					const uint64_t k0 = m_key[i];
					mine += k0 != KEY_EMPTY;
					if (part_ok && valid) {
						Entry<NW> e;
						e.key[0] = k0;
#pragma unroll
						for (int wv = 1; wv < NW; wv++)
							e.key[wv] = k0 == KEY_EMPTY ? KEY_EMPTY : m_key[wv * M + i];
						e.val = m_val[i];
						if constexpr (NW != 1) e.pad = 0;
						out.ent[sub0 + i] = e;
						out.aux[sub0 + i] = m_hi[i];
						if (TRACK) out.first[sub0 + i] = m_ord[i];
					}
					if (k0 != KEY_EMPTY) {
						m_key[i] = KEY_EMPTY;
						m_val[i] = 0;
						m_hi[i] = 0;
						if (TRACK) m_ord[i] = ORD_NONE;
					}
				}
#pragma unroll
				for (int dd = 32; dd > 0; dd >>= 1)
					mine += __shfl_down(mine, dd);
				if ((tid & 63) == 0 && mine)
					atomicAdd(&s_fill, mine);
				if (tid == 0)
					s_abort = 0;
				__syncthreads();
				const uint32_t d = s_fill;               // the keys of this part
				__syncthreads();
				if (tid == 0)
					s_fill = 0;
				BM_TICK(5);                              // image written + cleared
				// (a table must keep an empty slot: a look-up of an absent key ends there.  Past 15/16 the part is done again with room.)
				if (!part_ok || (uint64_t)d * 16 > (uint64_t)msz * 15) {
					ok = false;
					break;
				}
				total_d += d;
			}
			if (ok)
				break;
			// again with twice the room (more slots, or more parts once the image is as large as LDS allows).  The slots handed
			// out to the failed attempt are wiped: the scans of the table must find nothing in them.
			if (valid) {
				const uint64_t hi = (uint64_t)P * msz;
				for (uint64_t pos = (uint64_t)tid; pos < hi; pos += T)
					bm_store_empty<NW>(out, base + pos, TRACK);
			}
			if (msz < MEFF) msz = msz * 2 < MEFF ? msz * 2 : MEFF; else P *= 2;
			restarts++;
			BM_TICK(6);
		}
		if (tid == 0) {
			out.dir[u.dirix] = valid ? BmDir{base, msz, P} : BmDir{0, 0, 0};
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
