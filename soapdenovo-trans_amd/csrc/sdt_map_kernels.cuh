// sdt_map_kernels.cuh -- second pass over the reads on the GPU: prlRead2edge (prlRead2path.c:817-1335).
// Included by sdt_gpu.hip.
//
// After the host has cleaned the graph and built the edges it sends back, per node, one "path word":
//     bit 0      skip     = deleted || (linear && !inEdge)              (prlRead2path.c:650)
//     bit 1      linear
//     bits 2..3  twin
//     bits 32..  l_links  = edge id (interior nodes of an edge, node2edge.c:493-519)
// which k_set_paths writes over `val` of the node's table entry (the counts are no longer needed), plus the
// patch table of (K+1)-mers -> length-1 edges (node2edge.c:404-463), built on the host with the same hash.
//
// k_map_reads: ONE LANE PER READ walks its k-mers with rolling forward / reverse words (O(1) per k-mer, as the
// reference's chop does) and runs parse1read's little state machine (:617-789) in registers.  Half a million
// reads are in flight, so the dependent table lookups of one read are hidden behind the other reads; the bound
// is the random 16-byte load rate (~50 G/s), not latency.  Arcs go to a device hash map keyed (from << 32 | to):
// multiplicity by atomicAdd, first appearance (read ordinal << 16 | item index) by atomicMin -- that ordinal
// reproduces the reference's list order (new arcs pushed at the head, :427-428) without depending on the order
// in which reads are processed.
#pragma once

constexpr uint64_t PATH_SKIP = 1, PATH_LINEAR = 2;

template <int NW> struct PatchEnt {       // open addressing, key[0] == KEY_EMPTY marks a free slot
	uint64_t key[NW];
	uint64_t info;                        // edge id | twin << 32
};

struct ArcEnt {
	unsigned long long key;               // from << 32 | to ; 0 = empty (edge ids start at 1)
	unsigned long long first;
	unsigned int mult, pad;
};

template <int NW>
__global__ __launch_bounds__(TPB) void k_set_paths(Table<NW> tbl, const uint64_t *__restrict__ keys,
                                                   const uint64_t *__restrict__ info, uint64_t n, Stats *stats)
{
	uint32_t failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		Key<NW> k;
#pragma unroll
		for (int w = 0; w < NW; w++)
			k.w[w] = keys[i * NW + w];
		uint64_t slot;
		if (table_find<NW>(tbl, k, slot))
			tbl.ent[slot].val = info[i];
		else
			failed++;
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

template <int NW> __device__ inline bool lookup_path(const Table<NW> &tbl, const Key<NW> &k, uint64_t &info)
{
	uint64_t slot, lo, n;
	probe_begin<NW>(tbl, k, slot, lo, n);
	for (uint64_t probe = 0; probe < n; probe++, slot = probe_next(slot, lo, n)) {
		const Entry<NW> *e = tbl.ent + slot;
		if (NW == 1) {
			const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(e);
			if (kv.x == k.w[0]) { info = kv.y; return true; }
			if (kv.x == KEY_EMPTY) return false;
		} else {
			if (e->key[0] == KEY_EMPTY) return false;
			bool same = true;
#pragma unroll
			for (int w = 0; w < NW; w++)
				same = same && e->key[w] == k.w[w];
			if (same) { info = e->val; return true; }
		}
	}
	return false;
}

template <int NW> __device__ inline uint64_t lookup_patch(const PatchEnt<NW> *__restrict__ pt, uint64_t pmask, const Key<NW> &k)
{
	if (!pt)
		return 0;
	uint64_t slot = key_hash<NW>(k) & pmask;
	for (uint64_t probe = 0; probe <= pmask; probe++, slot = (slot + 1) & pmask) {
		const PatchEnt<NW> *e = pt + slot;
		if (e->key[0] == KEY_EMPTY)
			return 0;
		bool same = true;
#pragma unroll
		for (int w = 0; w < NW; w++)
			same = same && e->key[w] == k.w[w];
		if (same)
			return e->info;
	}
	return 0;
}

__device__ inline void arc_add(ArcEnt *arcs, uint64_t amask, uint32_t from, uint32_t to, uint64_t ord, uint32_t &failed)
{
	const unsigned long long key = ((unsigned long long)from << 32) | to;
	uint64_t slot = mix64(key) & amask;
	for (uint64_t probe = 0; probe < 4096; probe++, slot = (slot + 1) & amask) {
		unsigned long long k = __hip_atomic_load(&arcs[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (k == 0) {
			const unsigned long long old = atomicCAS(&arcs[slot].key, 0ULL, key);
			k = old == 0 ? key : old;
		}
		if (k == key) {
			atomicAdd(&arcs[slot].mult, 1u);
			atomicMin(&arcs[slot].first, (unsigned long long)ord);
			return;
		}
	}
	failed++;
}

// shift the NW-word value left by one base and append b (no mask)
template <int NW> __device__ inline Key<NW> key_revcomp_kplus1(const Key<NW> &plus, int K)
{
	if (K + 1 < 128) return key_revcomp<NW>(plus, K + 1);
	Key<NW> r = plus;
	r.w[NW - 1] = rev2bit(plus.w[NW - 1] ^ 0xAAAAAAAAAAAAAAAAULL);
	return r;
}


template <int NW>
__global__ __launch_bounds__(TPB) void k_map_reads(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ offs,
                                                   uint64_t nreads, int K, Table<NW> tbl, const PatchEnt<NW> *__restrict__ patch,
                                                   uint64_t pmask, ArcEnt *arcs, uint64_t amask, uint64_t ord_base,
                                                   uint64_t ord_stride, Stats *stats)
{
	uint32_t failed = 0, missing = 0;
	// masks for the rolling words
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	const int topbit = 2 * (K - 1);                  // where the reverse strand takes its new base
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r < nreads; r += (uint64_t)gridDim.x * TPB) {
		const uint64_t b0 = offs[r];
		const int len = (int)(offs[r + 1] - b0);
		if (len < K + 1)
			continue;                                // prlRead2path.c:969,1052,1116,1196
		const uint64_t ordinal = ord_base + r * ord_stride;
		Key<NW> fw, rc, prev_kmer;
#pragma unroll
		for (int i = 0; i < NW; i++)
			fw.w[i] = rc.w[i] = prev_kmer.w[i] = 0;
		int retain = 0, items = 0;
		bool have_prev = false, dead = false;
		uint64_t last_item = 0;
		uint32_t word = 0;
		for (int p = 0; p < len; p++) {
			const uint64_t bi = b0 + (uint64_t)p;
			if (p == 0 || (bi & 15) == 0)
				word = packed[bi >> 4];
			const uint32_t b = (word >> (30 - 2 * (int)(bi & 15))) & 3u;
			// forward: shift in at the low end; reverse complement: shift in the complement at the high end
			fw = key_append<NW>(fw, b);
#pragma unroll
			for (int i = 0; i < NW; i++)
				fw.w[i] &= mask.w[i];
#pragma unroll
			for (int i = NW - 1; i >= 0; i--)
				rc.w[i] = (rc.w[i] >> 2) | (i > 0 ? rc.w[i - 1] << 62 : 0ULL);
			{
				const int wi = NW - 1 - (topbit >> 6);
#pragma unroll
				for (int i = 0; i < NW; i++)
					if (i == wi)
						rc.w[i] |= (uint64_t)(b ^ 2u) << (topbit & 63);
			}
			if (p < K - 1)
				continue;
			const bool smaller = key_less<NW>(fw, rc);
			uint64_t info = 0;
			if (!lookup_path<NW>(tbl, smaller ? fw : rc, info)) {
				missing++;                           // "searchKmer: kmer ... is not found": cannot happen on the same reads
				break;
			}
			if (info & PATH_SKIP) {
				if (retain < 2) { retain = 0; items = 0; }
				else break;
				continue;
			}
			uint64_t item;
			bool append = false;
			if (info & PATH_LINEAR) {
				const uint64_t id = info >> 32, twin = (info >> 2) & 3u;
				item = smaller ? id : id + twin - 1;
				if (retain == 0 || have_prev) {
					append = true;
					have_prev = false;
				} else if (item != last_item) {
					append = true;
				}
			} else {
				if (have_prev) {
					// (K+1)-mer = previous vertex k-mer (as read) + last base of this one; canonical over K+1
					Key<NW> plus = key_append<NW>(prev_kmer, b);
					// K = 127: the reference's reverse complement of a 128-mer touches the last word only (host/graph/kw.h)
					Key<NW> bal = key_revcomp_kplus1<NW>(plus, K);
					const bool ps = key_less<NW>(plus, bal);
					const uint64_t pi = lookup_patch<NW>(patch, pmask, ps ? plus : bal);
					const uint64_t id = pi & 0xFFFFFFFFULL, twin = (pi >> 32) & 3u;
					item = pi == 0 ? 0 : (ps ? id : id + twin - 1);
					append = true;
				}
				have_prev = true;
				prev_kmer = fw;
				if (!append)
					continue;
			}
			if (append) {
				if (items >= 1 && !dead) {
					if (last_item == 0 || item == 0)
						dead = true;                 // signal 6 stops at the first unresolved item (:190-241)
					else
						arc_add(arcs, amask, (uint32_t)last_item, (uint32_t)item, (ordinal << 16) | (uint64_t)(items - 1), failed);
				}
				last_item = item;
				items++;
				retain++;
			}
		}
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
	if (missing)
		atomicAdd(&stats->scratch, (unsigned long long)missing);
}

static __global__ __launch_bounds__(TPB) void k_export_arcs(const ArcEnt *__restrict__ arcs, uint64_t slots, uint32_t *__restrict__ from,
                                                     uint32_t *__restrict__ to, uint32_t *__restrict__ mult,
                                                     uint64_t *__restrict__ first, unsigned long long max_n, unsigned long long *cursor)
{
	// eight slots per lane, one reservation per workgroup (sdt_append.cuh; max_n = 0: count only)
	__shared__ unsigned long long s_res[1 + TPB / 64];
	constexpr int IT = 8;
	for (uint64_t base = blockIdx.x * (uint64_t)TPB * IT; base < slots; base += (uint64_t)gridDim.x * TPB * IT) {
		uint32_t occ = 0;
#pragma unroll
		for (int j = 0; j < IT; j++) {
			const uint64_t s = base + (uint64_t)j * TPB + threadIdx.x;
			if (s < slots && arcs[s].key != 0) occ |= 1u << j;
		}
		unsigned long long pos = ap_block_reserve(__popc(occ), cursor, s_res);
		for (int j = 0; j < IT; j++) {
			if (!((occ >> j) & 1u)) continue;
			const unsigned long long at = pos++;
			if (at >= max_n) continue;
			const ArcEnt e = arcs[base + (uint64_t)j * TPB + threadIdx.x];
			from[at] = (uint32_t)(e.key >> 32);
			to[at] = (uint32_t)e.key;
			mult[at] = e.mult;
			first[at] = e.first;
		}
	}
}


// k_set_paths when the device already knows each node's host index (sdt_gpu_set_node_index): no keys, no probing
template <int NW>
__global__ __launch_bounds__(TPB) void k_set_paths_by_index(Table<NW> tbl, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ info, uint64_t n,
                                                            Stats *stats)
{
	const uint64_t slots = tbl.slots();
	uint32_t failed = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY) continue;
		const uint64_t i = idx[s];
		if (i < n) tbl.ent[s].val = info[i];
		else failed++;
	}
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}
