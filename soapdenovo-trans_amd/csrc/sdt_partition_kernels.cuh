// sdt_partition_kernels.cuh -- kernels of the locality pipeline (design notes: sdt_partition.cuh).
// Included by sdt_gpu.hip after stage_tile / TileView / chop_record are defined.
#pragma once

// ---- k_part_hist: exact sizes of the final buckets ---------------------------------------------------------
__global__ __launch_bounds__(PT_TPB) void k_part_hist(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                                      uint64_t nreads, int K, int max_tile_words, int tile_smem_words,
                                                      PartGeom geo, unsigned int *__restrict__ ghist)
{
	extern __shared__ uint32_t smem[];
	uint32_t *s_hist = smem + tile_smem_words;
	for (int i = threadIdx.x; i < NBF; i += PT_TPB)
		s_hist[i] = 0;
	__syncthreads();
	const uint64_t ntiles = (nreads + TILE_READS - 1) / TILE_READS;
	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const TileView tv = stage_tile(smem, max_tile_words, packed, offs, tile * TILE_READS, nreads, K);
		for (uint32_t q = threadIdx.x; q < tv.nk; q += PT_TPB) {
			const int r = tile_find_read(tv.pre, q);
			const int j = (int)(q - tv.pre[r]);
			const int len = (int)(tv.rb[r + 1] - tv.rb[r]);
			uint32_t prev, next;
			const Key<1> key = chop_record<1>(tv.words, (int)tv.rb[r], len, j, K, prev, next);
			const uint64_t h = bij_fwd(key.w[0], geo.n);
			atomicAdd(&s_hist[(uint32_t)(h >> geo.tagbits)], 1u);
		}
		__syncthreads();
	}
	__syncthreads();
	for (int i = threadIdx.x; i < NBF; i += PT_TPB)
		if (s_hist[i])
			atomicAdd(&ghist[i], s_hist[i]);
}

// ---- k_part_scan: offsets, cursors, tile table (one workgroup of 1024) -------------------------------------
__global__ __launch_bounds__(1024) void k_part_scan(PartBufs pb)
{
	__shared__ unsigned long long s_sum[1024];
	__shared__ unsigned int s_tiles[NB1 + 1];
	constexpr int PER = NBF / 1024;
	const int t = threadIdx.x;
	unsigned long long loc[PER], sum = 0;
#pragma unroll
	for (int i = 0; i < PER; i++) {
		loc[i] = sum;
		sum += pb.hist[t * PER + i];
	}
	s_sum[t] = sum;
	__syncthreads();
	// Hillis-Steele inclusive scan over 1024 partial sums
	for (int d = 1; d < 1024; d <<= 1) {
		unsigned long long v = t >= d ? s_sum[t - d] : 0;
		__syncthreads();
		s_sum[t] += v;
		__syncthreads();
	}
	const unsigned long long base = t ? s_sum[t - 1] : 0;
#pragma unroll
	for (int i = 0; i < PER; i++) {
		const int f = t * PER + i;
		pb.off2[f] = base + loc[i];
		pb.cursor2[f] = base + loc[i];
	}
	if (t == 1023)
		pb.off2[NBF] = s_sum[1023];
	__syncthreads();
	// start of L1 bucket t = off2[t * NB2]; that element is the first one of thread t * NB2 / PER, whose
	// exclusive base is the inclusive sum of the thread before it (kept in LDS: no global re-read)
	if (t < NB1) {
		constexpr int STEP = NB2 / PER;
		static_assert(NB2 % PER == 0, "an L1 bucket must start on a thread boundary of the scan");
		const unsigned long long b0 = t ? s_sum[t * STEP - 1] : 0ULL;
		const unsigned long long b1 = s_sum[(t + 1) * STEP - 1];
		pb.cursor1[t] = b0;
		s_tiles[t] = (unsigned int)((b1 - b0 + L2_TILE - 1) / L2_TILE);
	}
	__syncthreads();
	if (t == 0) {
		unsigned int acc = 0;
		for (int b = 0; b < NB1; b++) {
			const unsigned int n = s_tiles[b];
			pb.tile1[b] = acc;
			acc += n;
		}
		pb.tile1[NB1] = acc;
	}
}

// ---- k_part_l1: chop + scatter into the L1 buckets of A ----------------------------------------------------
__global__ __launch_bounds__(PT_TPB) void k_part_l1(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                                    uint64_t nreads, int K, int max_tile_words, int tile_smem_words,
                                                    PartGeom geo, PartBufs pb)
{
	extern __shared__ uint32_t smem[];
	uint32_t *s_cnt = smem + tile_smem_words;                       // NB1
	uint32_t *s_fill = s_cnt + NB1;                                 // NB1
	unsigned long long *s_base = (unsigned long long *)(s_fill + NB1);   // NB1 (8-byte aligned: tile_smem_words is even)
	const uint64_t ntiles = (nreads + TILE_READS - 1) / TILE_READS;
	const int keep = geo.n - L1BITS;
	for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const TileView tv = stage_tile(smem, max_tile_words, packed, offs, tile * TILE_READS, nreads, K);
		if (threadIdx.x < NB1) {
			s_cnt[threadIdx.x] = 0;
			s_fill[threadIdx.x] = 0;
		}
		__syncthreads();
		for (uint32_t q = threadIdx.x; q < tv.nk; q += PT_TPB) {
			const int r = tile_find_read(tv.pre, q);
			const int j = (int)(q - tv.pre[r]);
			const int len = (int)(tv.rb[r + 1] - tv.rb[r]);
			uint32_t prev, next;
			const Key<1> key = chop_record<1>(tv.words, (int)tv.rb[r], len, j, K, prev, next);
			const uint64_t h = bij_fwd(key.w[0], geo.n);
			atomicAdd(&s_cnt[(uint32_t)(h >> keep)], 1u);
		}
		__syncthreads();
		if (threadIdx.x < NB1) {
			const uint32_t n = s_cnt[threadIdx.x];
			s_base[threadIdx.x] = n ? atomicAdd(&pb.cursor1[threadIdx.x], (unsigned long long)n) : 0ULL;
		}
		__syncthreads();
		for (uint32_t q = threadIdx.x; q < tv.nk; q += PT_TPB) {
			const int r = tile_find_read(tv.pre, q);
			const int j = (int)(q - tv.pre[r]);
			const int len = (int)(tv.rb[r + 1] - tv.rb[r]);
			uint32_t prev, next;
			const Key<1> key = chop_record<1>(tv.words, (int)tv.rb[r], len, j, K, prev, next);
			const uint64_t h = bij_fwd(key.w[0], geo.n);
			const uint32_t b = (uint32_t)(h >> keep);
			const uint32_t pos = atomicAdd(&s_fill[b], 1u);
			pb.A[s_base[b] + pos] = make_record(h, prev, next, keep);
		}
		__syncthreads();
	}
}

// ---- k_part_l2: A -> B, each L1 bucket split into its NB2 sub-buckets --------------------------------------
__global__ __launch_bounds__(PT_TPB) void k_part_l2(PartGeom geo, PartBufs pb)
{
	__shared__ uint32_t s_cnt[NB2], s_fill[NB2];
	__shared__ unsigned long long s_base[NB2];
	__shared__ unsigned int s_tile1[NB1 + 1];
	for (int i = threadIdx.x; i <= NB1; i += PT_TPB)
		s_tile1[i] = pb.tile1[i];
	__syncthreads();
	const unsigned int ntiles = s_tile1[NB1];
	const int l2shift = geo.tagbits + 5;
	const uint64_t outmask = (1ULL << l2shift) - 1;
	for (unsigned int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		// which L1 bucket: largest b with tile1[b] <= tile
		int lo = 0, hi = NB1;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (s_tile1[mid] <= tile) lo = mid; else hi = mid;
		}
		const int b1 = lo;
		const unsigned long long bstart = pb.off2[(size_t)b1 * NB2], bend = pb.off2[(size_t)(b1 + 1) * NB2];
		const unsigned long long t0 = bstart + (unsigned long long)(tile - s_tile1[b1]) * L2_TILE;
		const unsigned int n = (unsigned int)((bend - t0) < (unsigned long long)L2_TILE ? (bend - t0) : (unsigned long long)L2_TILE);
		if (threadIdx.x < NB2) {
			s_cnt[threadIdx.x] = 0;
			s_fill[threadIdx.x] = 0;
		}
		__syncthreads();
		for (unsigned int i = threadIdx.x; i < n; i += PT_TPB)
			atomicAdd(&s_cnt[(uint32_t)(pb.A[t0 + i] >> l2shift)], 1u);
		__syncthreads();
		if (threadIdx.x < NB2) {
			const uint32_t c = s_cnt[threadIdx.x];
			s_base[threadIdx.x] = c ? atomicAdd(&pb.cursor2[(size_t)b1 * NB2 + threadIdx.x], (unsigned long long)c) : 0ULL;
		}
		__syncthreads();
		for (unsigned int i = threadIdx.x; i < n; i += PT_TPB) {
			const uint64_t rec = pb.A[t0 + i];
			const uint32_t l2 = (uint32_t)(rec >> l2shift);
			const uint32_t pos = atomicAdd(&s_fill[l2], 1u);
			pb.B[s_base[l2] + pos] = rec & outmask;
		}
		__syncthreads();
	}
}

// ---- k_part_final: one workgroup per final bucket, count in LDS, merge once into the node table ------------
__global__ __launch_bounds__(FIN_TPB) void k_part_final(PartGeom geo, PartBufs pb, unsigned int first_bucket, Table<1> tbl,
                                                        Stats *stats)
{
	__shared__ uint64_t s_tag[FIN_SLOTS];
	__shared__ uint64_t s_val[FIN_SLOTS];
	__shared__ unsigned int s_fillcnt, s_ndiv;
	const unsigned int f = first_bucket + blockIdx.x;
	const unsigned long long start = pb.off2[f], end = pb.off2[f + 1];
	if (start == end)
		return;
	for (int i = threadIdx.x; i < FIN_SLOTS; i += FIN_TPB) {
		s_tag[i] = ~0ULL;
		s_val[i] = 0;
	}
	if (threadIdx.x == 0) {
		s_fillcnt = 0;
		s_ndiv = 0;
	}
	__syncthreads();
	const uint64_t hbase = (uint64_t)f << geo.tagbits;
	uint32_t claimed = 0, failed = 0;
	for (unsigned long long i = start + threadIdx.x; i < end; i += FIN_TPB) {
		const uint64_t rec = pb.B[i];
		const uint64_t tag = rec >> 5;
		const uint32_t nb = (uint32_t)rec & 31u;
		const uint32_t prev = nb / 5u, next = nb - prev * 5u;
		uint32_t s = (uint32_t)((tag * 0x9E3779B97F4A7C15ULL) >> 52) & (FIN_SLOTS - 1);
		bool done = false;
		for (int probe = 0; probe < 48 && !done; probe++, s = (s + 1) & (FIN_SLOTS - 1)) {
			uint64_t t = s_tag[s];
			if (t == ~0ULL) {
				if (__hip_atomic_load(&s_fillcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)FIN_MAX_FILL)
					break;                                   // table full enough: this record goes to the global table
				const uint64_t old = atomicCAS((unsigned long long *)&s_tag[s], ~0ULL, (unsigned long long)tag);
				if (old == ~0ULL) {
					atomicAdd(&s_fillcnt, 1u);
					t = tag;
				} else {
					t = old;
				}
			}
			if (t != tag)
				continue;
			// saturating update of the LDS word, same layout as the global `val`
			uint64_t seen = s_val[s];
			const int ls = 6 * (int)prev, rs = 24 + 6 * (int)next;
			for (;;) {
				if ((seen >> 48) >= 0xFFFFu - 2u * FIN_TPB)
					break;                                   // 16-bit count nearly full: divert (done stays false)
				const bool l_done = prev >= 4u || ((seen >> ls) & 63u) >= 63u;
				const bool r_done = next >= 4u || ((seen >> rs) & 63u) >= 63u;
				if (l_done && r_done) {
					atomicAdd((unsigned long long *)&s_val[s], (unsigned long long)VAL_COUNT_ONE);
					done = true;
					break;
				}
				uint64_t nv = seen + VAL_COUNT_ONE;
				if (!l_done) nv += 1ULL << ls;
				if (!r_done) nv += 1ULL << rs;
				const uint64_t got = atomicCAS((unsigned long long *)&s_val[s], (unsigned long long)seen, (unsigned long long)nv);
				if (got == seen) {
					done = true;
					break;
				}
				seen = got;
			}
			break;                                           // found our slot: either counted or diverted
		}
		if (!done) {
			// Not countable in LDS (table full / count field nearly full).  Doing the global insert right here
			// would make every wave iteration wait for a chain of dependent memory-side atomics with most lanes
			// idle; park the record instead (buffer A is free after k_part_l2 and has the same bucket ranges)
			// and insert all parked records afterwards with every lane busy.
			const unsigned int slot = atomicAdd(&s_ndiv, 1u);
			pb.A[start + slot] = rec;
		}
	}
	__syncthreads();
	const unsigned int ndiv = s_ndiv;
	for (unsigned int i = threadIdx.x; i < ndiv; i += FIN_TPB) {
		const uint64_t rec = __hip_atomic_load(&pb.A[start + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t nb = (uint32_t)rec & 31u;
		const uint32_t prev = nb / 5u, next = nb - prev * 5u;
		Key<1> key{{bij_inv(hbase | (rec >> 5), geo.n)}};
		if (!table_put<1>(tbl, key, prev, next, claimed))
			failed++;
	}
	__syncthreads();
	for (int i = threadIdx.x; i < FIN_SLOTS; i += FIN_TPB) {
		const uint64_t tag = s_tag[i];
		if (tag == ~0ULL)
			continue;
		const uint64_t add = s_val[i];
		if ((add >> 48) == 0)
			continue;                                        // claimed, but every occurrence was diverted
		if (!table_merge(tbl, bij_inv(hbase | tag, geo.n), add, claimed))
			failed++;
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) {
		claimed += __shfl_down(claimed, d);
		failed += __shfl_down(failed, d);
	}
	if ((threadIdx.x & 63) == 0) {
		if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
		if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
	}
	if (threadIdx.x == 0)
		atomicAdd(&stats->kmers, end - start);
}
