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
	static constexpr int T = 512;                                                      // two workgroups per CU
	static constexpr int SLOT_BYTES = NW * 8 + 8 + 4 + (TRACK ? 8 : 0);
	static constexpr int M = (72 * 1024 / SLOT_BYTES) / 64 * 64;                       // LDS merge table: 3648 / 2624 / 2624 / 2048 / 1664 / 1408 slots
	static constexpr uint32_t CAP = (uint32_t)M * 3u / 4u;                             // keys it takes
	static constexpr int BITW = M / 16;                                                // bitmap of a table being written: 2 M slots at most
	static constexpr size_t SMEM = (size_t)M * SLOT_BYTES + (size_t)BITW * 4;
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
enum { BM_CTL_SLOTS, BM_CTL_NODES, BM_CTL_RESTARTS, BM_CTL_MAXPARTS, BM_CTL_N };

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

// size class of a bucket's input (position of the highest bit of entries + nodes), 0 = nothing
__device__ inline uint32_t bm_class_of(const unsigned long long *dpre, const uint32_t *old_cnt, const uint32_t *xoff, uint32_t b)
{
	unsigned long long n = dpre[b + 1] - dpre[b];
	if (old_cnt) n += old_cnt[b];
	if (xoff) n += xoff[b + 1] - xoff[b];
	return n ? 64u - (uint32_t)__clzll((long long)n) : 0u;
}

__global__ __launch_bounds__(256) void k_bm_class_hist(const unsigned long long *__restrict__ dpre, const uint32_t *__restrict__ old_cnt, const uint32_t *__restrict__ xoff,
                                                       uint32_t nb, uint32_t *__restrict__ ccnt)
{
	__shared__ uint32_t s_c[65];
	if (threadIdx.x < 65) s_c[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nb; b += gridDim.x * 256u)
		atomicAdd(&s_c[bm_class_of(dpre, old_cnt, xoff, b)], 1u);
	__syncthreads();
	if (threadIdx.x < 65 && s_c[threadIdx.x])
		atomicAdd(&ccnt[threadIdx.x], s_c[threadIdx.x]);
}

// (one thread) start of every class in the order, largest class first; cfill = 0
__global__ void k_bm_class_scan(const uint32_t *__restrict__ ccnt, uint32_t *__restrict__ cstart, uint32_t *__restrict__ cfill)
{
	uint32_t acc = 0;
	for (int c = 64; c >= 0; c--) {
		cstart[c] = acc;
		cfill[c] = 0;
		acc += ccnt[c];
	}
}

__global__ __launch_bounds__(256) void k_bm_class_place(const unsigned long long *__restrict__ dpre, const uint32_t *__restrict__ old_cnt, const uint32_t *__restrict__ xoff,
                                                        uint32_t nb, const uint32_t *__restrict__ cstart, uint32_t *__restrict__ cfill, uint32_t *__restrict__ order)
{
	for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nb; b += gridDim.x * 256u) {
		const uint32_t c = bm_class_of(dpre, old_cnt, xoff, b);
		order[cstart[c] + atomicAdd(&cfill[c], 1u)] = b;
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
		uint64_t nv = 0;
#pragma unroll
		for (int f = 0; f < 8; f++) {
			const uint32_t a = (uint32_t)(seen >> (6 * f)) & 63u, b = (uint32_t)(add >> (6 * f)) & 63u;
			const uint32_t s = a + b > 63u ? 63u : a + b;
			nv |= (uint64_t)s << (6 * f);
		}
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

// one entry into the LDS merge table (pass p of P takes the keys whose hash falls into part p); no room: *abort = 1
template <int NW, bool TRACK, int M>
__device__ __forceinline__ void bm_insert(unsigned long long *m_key, unsigned long long *m_val, unsigned long long *m_ord, uint32_t *m_hi, uint32_t *fill, uint32_t *abort_flag,
                                 uint32_t cap, uint32_t P, uint32_t p, const Key<NW> &key, uint64_t val, uint32_t auxadd, uint64_t ord)
{
	if (P > 1 && __umulhi((uint32_t)(key_hash<NW>(key) >> 32), P) != p)
		return;
	const int s = sk_lds_locate<NW, M>(m_key, fill, key, cap);
	if (s < 0) {
		*abort_flag = 1;
		return;
	}
	bm_lds_merge(&m_val[s], &m_hi[s], val, auxadd);
	if (TRACK && ord != ORD_NONE)
		atomicMin(&m_ord[s], (unsigned long long)ord);
}

template <int NW, bool TRACK>
__global__ __launch_bounds__((BmGeo<NW, TRACK>::T)) void k_bm_finalize(BmIn<NW> in, BmOut<NW> out, const uint32_t *__restrict__ order, uint32_t nb,
                                                                      uint32_t *__restrict__ next, int K, Stats *stats)
{
	using G = BmGeo<NW, TRACK>;
	constexpr int M = G::M, T = G::T, BITW = G::BITW;
	constexpr uint32_t CAP = G::CAP;
	constexpr int LW = NW + 1 + (TRACK ? 1 : 0), XW = BmX<NW>::W;
	extern __shared__ unsigned long long bm_sm[];
	unsigned long long *m_key = bm_sm;                               // NW x M, word-major (sk_lds_locate's layout)
	unsigned long long *m_val = m_key + NW * M;                      // M
	unsigned long long *m_ord = m_val + M;                           // M when TRACK
	uint32_t *m_hi = (uint32_t *)(m_ord + (TRACK ? M : 0));          // M: the node's aux word
	uint32_t *s_bits = m_hi + M;                                     // BITW
	__shared__ uint32_t s_fill, s_abort, s_b;
	__shared__ unsigned long long s_base;
	const int tid = threadIdx.x;
	uint32_t restarts = 0, maxparts = 0;
	unsigned long long nodes = 0;
	for (;;) {
		if (tid == 0)
			s_b = atomicAdd(next, 1u);
		__syncthreads();
		const uint32_t at = s_b;
		__syncthreads();
		if (at >= nb)
			break;
		const uint32_t b = order[at];
		const uint32_t d0 = in.doff[b], d1 = in.doff[b + 1];
		const unsigned long long nlog = in.dpre[b + 1] - in.dpre[b];
		BmDir od = {0, 0, 0};
		uint32_t nold = 0;
		if (in.old.dir) {
			od = in.old.dir[b];
			nold = in.old_cnt[b];
		}
		const uint32_t x0 = in.xent ? in.xoff[b] : 0u, x1 = in.xent ? in.xoff[b + 1] : 0u;
		const unsigned long long n_in = nlog + nold + (x1 - x0);
		if (n_in == 0) {
			if (tid == 0) {
				out.dir[b] = BmDir{0, 0, 0};
				out.cnt[b] = 0;
			}
			continue;
		}
		// parts: one when everything fits whatever the keys are; else from a guess of the distinct keys (an entry of the log is one of
		// ~2.5 of its key, a node of a table is the only one), doubled whenever the LDS table fills up all the same
		uint32_t P = 1;
		if (n_in > CAP) {
			const unsigned long long est = nold + (x1 - x0) + nlog / 2 + 1;
			const unsigned long long per = (unsigned long long)CAP * 85 / 100;
			P = (uint32_t)((est + per - 1) / per);
		}
		uint32_t margin = 0, floor_ssub = 8;
		unsigned long long base = ~0ULL;
		uint32_t ssub = 0, total_d = 0;
		for (;;) {
			bool ok = true, lds_full = false;
			base = ~0ULL;
			ssub = 0;
			total_d = 0;
			bool allocated = false;
			for (uint32_t p = 0; p < P; p++) {
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
				// sources: the bucket's segments of the log, then its nodes in the table of an earlier finalize, then its nodes out of
				// the flat table -- one loop, one place where an entry goes into the LDS table
				const uint32_t nsrc = (d1 - d0) + 2u;
				for (uint32_t src = 0; src < nsrc; src++) {
					const int kind = src < d1 - d0 ? 0 : (src == d1 - d0 ? 1 : 2);
					const uint64_t *seg = nullptr;
					uint64_t cnt = 0;
					if (kind == 0) {
						const LogDesc ds = in.desc[d0 + src];
						seg = (const uint64_t *)ds.ptr;
						cnt = ds.count;
					} else if (kind == 1) {
						cnt = od.parts ? (uint64_t)od.parts * od.ssub : 0;
					} else {
						cnt = x1 - x0;
					}
					for (uint64_t e = (uint64_t)tid; e < cnt; e += T) {
						Key<NW> key;
						uint64_t val, ord = ORD_NONE;
						uint32_t auxadd = 0;
						if (kind == 0) {
							const uint64_t *x = seg + (size_t)e * LW;
							if (NW == 1 && !TRACK) {
								const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(x);
								key.w[0] = kv.x;
								val = kv.y;
							} else {
#pragma unroll
								for (int i = 0; i < NW; i++)
									key.w[i] = x[i];
								val = x[NW];
								if (TRACK) ord = x[NW + (TRACK ? 1 : 0)];
							}
						} else if (kind == 1) {
							const uint64_t sl = od.base + e;
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
						bm_insert<NW, TRACK, M>(m_key, m_val, m_ord, m_hi, &s_fill, &s_abort, CAP, P, p, key, val, auxadd, ord);
					}
				}
				__syncthreads();
				const uint32_t d = s_fill;
				if (s_abort) {
					ok = false;
					lds_full = true;
					break;
				}
				if (p == 0) {
					// slots per table: load 3/4; with several parts a quarter (and more after a failed attempt) on top of what part 0 holds
					uint64_t want = P == 1 ? ((uint64_t)d * 4 + 2) / 3 + 1 : ((uint64_t)d * (5 + margin) / 4 * 4 + 2) / 3 + 16;
					want = (want + 7) & ~7ULL;
					if (want < floor_ssub) want = floor_ssub;
					if (want > (uint64_t)BITW * 32) want = (uint64_t)BITW * 32;
					ssub = (uint32_t)want;
					if (tid == 0) {
						const unsigned long long need = (unsigned long long)P * ssub;
						const unsigned long long got = atomicAdd(&out.ctl[BM_CTL_SLOTS], need);
						s_base = got + need <= out.cap ? got : ~0ULL;
					}
					__syncthreads();
					base = s_base;
					allocated = base != ~0ULL;
				}
				if ((uint64_t)d * 16 > (uint64_t)ssub * 15) {     // (one empty slot at the very least: a look-up of an absent key must end)
					ok = false;
					floor_ssub = (uint32_t)((((uint64_t)d * 5 / 3 + 16) + 7) & ~7ULL);      // what this part needs, and a quarter
					break;
				}
				total_d += d;
				if (base != ~0ULL) {
					// write the table of this part: every node claims the first free slot from its home on, in a bitmap
					const uint64_t sub0 = base + (uint64_t)p * ssub;
					for (int i = tid; i < BITW; i += T)
						s_bits[i] = 0;
					__syncthreads();
					for (int i = tid; i < M; i += T) {
						if (m_key[i] == KEY_EMPTY)
							continue;
						Key<NW> key;
						key.w[0] = m_key[i];
#pragma unroll
						for (int wv = 1; wv < NW; wv++)
							key.w[wv] = m_key[wv * M + i];
						uint32_t pos = __umulhi((uint32_t)key_hash<NW>(key), ssub);
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
					for (uint32_t pos = (uint32_t)tid; pos < ssub; pos += T) {
						if (s_bits[pos >> 5] & (1u << (pos & 31u)))
							continue;
						Entry<NW> e;
#pragma unroll
						for (int wv = 0; wv < NW; wv++)
							e.key[wv] = KEY_EMPTY;
						e.val = 0;
						if constexpr (NW != 1) e.pad = 0;
						out.ent[sub0 + pos] = e;
						out.aux[sub0 + pos] = 0;
						if (TRACK) out.first[sub0 + pos] = ORD_NONE;
					}
				}
				__syncthreads();                             // (the LDS table is cleared for the next part)
			}
			if (ok)
				break;
			// again, with more parts (the LDS table filled up) or more room per part (a part turned out larger than part 0 suggested).
			// Slots handed out to the failed attempt are wiped: the scans of the table must find nothing in them.
			if (allocated) {
				const uint64_t hi = (uint64_t)P * ssub;
				for (uint64_t pos = (uint64_t)tid; pos < hi; pos += T) {
					Entry<NW> e;
#pragma unroll
					for (int wv = 0; wv < NW; wv++)
						e.key[wv] = KEY_EMPTY;
					e.val = 0;
					if constexpr (NW != 1) e.pad = 0;
					out.ent[base + pos] = e;
					out.aux[base + pos] = 0;
					if (TRACK) out.first[base + pos] = ORD_NONE;
				}
			}
			__syncthreads();
			if (lds_full) P *= 2; else margin += 2;
			restarts++;
		}
		if (tid == 0) {
			out.dir[b] = base != ~0ULL ? BmDir{base, ssub, P} : BmDir{0, 0, 0};
			out.cnt[b] = total_d;
		}
		nodes += total_d;
		maxparts = P > maxparts ? P : maxparts;
	}
	if (tid == 0) {
		if (nodes) atomicAdd(&out.ctl[BM_CTL_NODES], nodes);
		if (restarts) atomicAdd(&out.ctl[BM_CTL_RESTARTS], (unsigned long long)restarts);
		atomicMax(&out.ctl[BM_CTL_MAXPARTS], (unsigned long long)maxparts);
	}
}

