// sdt_table_kernels.cuh -- kernels over the node table: the direct pass-1 family (k_count_reads: one device atomic per occurrence)
// and the scans (clear, -d filter, linear marks + kmerFreq bins, export / import, growth).  Reference call sites at every kernel.
#pragma once
#include "sdt_tile.cuh"

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
