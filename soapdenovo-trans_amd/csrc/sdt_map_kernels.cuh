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
		uint64_t slot = key_hash<NW>(k) & tbl.mask;
		bool ok = false;
		for (uint64_t probe = 0; probe <= tbl.mask; probe++, slot = (slot + 1) & tbl.mask) {
			const Entry<NW> *e = tbl.ent + slot;
			if (e->key[0] == KEY_EMPTY)
				break;
			bool same = true;
#pragma unroll
			for (int w = 0; w < NW; w++)
				same = same && e->key[w] == k.w[w];
			if (same) {
				tbl.ent[slot].val = info[i];
				ok = true;
				break;
			}
		}
		if (!ok)
			failed++;
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

template <int NW> __device__ inline bool lookup_path(const Table<NW> &tbl, const Key<NW> &k, uint64_t &info)
{
	uint64_t slot = key_hash<NW>(k) & tbl.mask;
	for (uint64_t probe = 0; probe <= tbl.mask; probe++, slot = (slot + 1) & tbl.mask) {
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

template <int NW> __device__ inline Key<NW> key_append(const Key<NW> &k, uint32_t b)
{
	Key<NW> r;
#pragma unroll
	for (int i = 0; i < NW; i++)
		r.w[i] = (k.w[i] << 2) | (i + 1 < NW ? k.w[i + 1] >> 62 : (uint64_t)b);
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

__global__ __launch_bounds__(TPB) void k_export_arcs(const ArcEnt *__restrict__ arcs, uint64_t slots, uint32_t *__restrict__ from,
                                                     uint32_t *__restrict__ to, uint32_t *__restrict__ mult,
                                                     uint64_t *__restrict__ first, unsigned long long max_n, unsigned long long *cursor)
{
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const ArcEnt e = arcs[s];
		if (e.key == 0)
			continue;
		const unsigned long long pos = atomicAdd(cursor, 1ULL);
		if (pos >= max_n)
			continue;
		from[pos] = (uint32_t)(e.key >> 32);
		to[pos] = (uint32_t)e.key;
		mult[pos] = e.mult;
		first[pos] = e.first;
	}
}


// ---------------------------------------------------------------------------------------------------------------
// Graph-cleaning dry runs on the device (cutTipPreGraph.c).  The host owns the ORDER (layout replay, ordered commit
// of the few visits that write); what it needs from a sweep is the read-only part -- the walks -- and those are
// table look-ups, which this chip does at tens of G/s.  The device table mirrors the host graph: the host sends
// back the nodes it wrote (k_update_nodes) -- the nodes its own
// Mark1in1outNode marked included -- and the index each node has in its visiting order (k_set_index).
// ---------------------------------------------------------------------------------------------------------------
template <int NW> __device__ inline bool find_slot(const Table<NW> &tbl, const Key<NW> &k, uint64_t &slot_out)
{
	uint64_t slot = key_hash<NW>(k) & tbl.mask;
	for (uint64_t probe = 0; probe <= tbl.mask; probe++, slot = (slot + 1) & tbl.mask) {
		const Entry<NW> *e = tbl.ent + slot;
		if (e->key[0] == KEY_EMPTY)
			return false;
		bool same = true;
#pragma unroll
		for (int w = 0; w < NW; w++)
			same = same && e->key[w] == k.w[w];
		if (same) {
			slot_out = slot;
			return true;
		}
	}
	return false;
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_set_index(Table<NW> tbl, const uint64_t *__restrict__ keys, uint64_t n,
                                                   uint64_t *__restrict__ idx, Stats *stats)
{
	uint32_t failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		Key<NW> k;
#pragma unroll
		for (int w = 0; w < NW; w++)
			k.w[w] = keys[i * NW + w];
		uint64_t slot;
		if (find_slot<NW>(tbl, k, slot)) idx[slot] = i;
		else failed++;
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

// links + flags of the given nodes as the host has them now (count is never changed by the cleaning passes)
template <int NW>
__global__ __launch_bounds__(TPB) void k_update_nodes(Table<NW> tbl, const uint64_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ l_links, const uint32_t *__restrict__ r_flags,
                                                      uint64_t n, Stats *stats)
{
	uint32_t failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		Key<NW> k;
#pragma unroll
		for (int w = 0; w < NW; w++)
			k.w[w] = keys[i * NW + w];
		uint64_t slot;
		if (!find_slot<NW>(tbl, k, slot)) { failed++; continue; }
		const uint64_t v = tbl.ent[slot].val;
		tbl.ent[slot].val = (v & 0xFFFF000000000000ULL) | ((uint64_t)(r_flags[i] & 0xFFFFFFu) << 24) | (uint64_t)(l_links[i] & 0xFFFFFFu);
		const uint32_t a = tbl.aux[slot] & 0xFFFFu;
		tbl.aux[slot] = a | ((r_flags[i] >> 24 & 1u) ? AUX_LINEAR : 0u) | ((r_flags[i] >> 25 & 1u) ? AUX_DELETED : 0u);
	}
	if (failed)
		atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

__device__ inline uint32_t dev_degree(uint64_t links24)
{
	uint32_t d = 0;
#pragma unroll
	for (int b = 0; b < 4; b++)
		d += ((links24 >> (6 * b)) & 63u) != 0;
	return d;
}
__device__ inline uint32_t first_link(uint64_t links24)
{
	uint32_t b = 0;
	while (b < 4 && ((links24 >> (6 * b)) & 63u) == 0) b++;
	return b;
}

template <int NW> __device__ inline Key<NW> key_next_masked(const Key<NW> &k, uint32_t b, const Key<NW> &mask)
{
	Key<NW> r = key_append<NW>(k, b);
#pragma unroll
	for (int i = 0; i < NW; i++)
		r.w[i] &= mask.w[i];
	return r;
}

// the walk of clipTipFromNode (cutTipPreGraph.c:43-281) from every node, read-only.  Output, at the HOST index of
// the node: end = host index of the node the walk stopped at (~0 = nothing to decide), info = ch | sm << 2 |
// thin_stop << 3 (the base by which the end node sees the chain, the strand on which it was reached).
template <int NW>
__global__ __launch_bounds__(TPB) void k_tip_walks(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, int thin, int cut_len,
                                                   uint64_t *__restrict__ end_out, uint8_t *__restrict__ info_out, Stats *stats,
                                                   uint64_t *__restrict__ rec = nullptr, unsigned long long max_rec = 0,
                                                   unsigned long long *cursor = nullptr)
{
	const uint64_t slots = tbl.mask + 1;
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	uint32_t missing = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		const uint64_t me = idx[s];
		if (end_out) {
			end_out[me] = ~0ULL;
			info_out[me] = 0;
		}
		const uint32_t a = tbl.aux[s];
		if (a & (AUX_LINEAR | AUX_DELETED)) continue;
		const bool single = (e.val >> 48) == 1 && (a & 0xFFFFu) == 0;
		if (thin && !single) continue;
		const uint64_t ll = e.val & 0xFFFFFFu, rl = (e.val >> 24) & 0xFFFFFFu;
		const uint32_t in = dev_degree(ll), out = dev_degree(rl);
		Key<NW> at;
#pragma unroll
		for (int w = 0; w < NW; w++) at.w[w] = e.key[w];
		uint32_t b;
		if (in == 0 && out == 1) {
			b = first_link(rl);
		} else if (in == 1 && out == 0) {
			at = key_revcomp<NW>(at, K);
			b = first_link(ll) ^ 2u;
		} else {
			continue;
		}
		int steps = 1;
		uint32_t thin_stop = 0;
		bool give_up = false;
		Key<NW> step = key_next_masked<NW>(at, b, mask);
		Key<NW> bal = key_revcomp<NW>(step, K);
		bool sm = !key_less<NW>(bal, step);               // KmerLarger(word, bal) -> take bal, smaller = 0
		uint64_t os;
		if (!find_slot<NW>(tbl, sm ? step : bal, os)) { missing++; continue; }
		for (;;) {
			const uint32_t oa = tbl.aux[os];
			if (!(oa & AUX_LINEAR)) break;
			steps++;
			const uint64_t ov = tbl.ent[os].val;
			if (thin && !((ov >> 48) == 1 && (oa & 0xFFFFu) == 0)) { thin_stop = 1; break; }
			if (steps > cut_len) { give_up = true; break; }
			at = step;
			b = sm ? first_link((ov >> 24) & 0xFFFFFFu) : (first_link(ov & 0xFFFFFFu) ^ 2u);
			step = key_next_masked<NW>(at, b, mask);
			bal = key_revcomp<NW>(step, K);
			sm = !key_less<NW>(bal, step);
			if (!find_slot<NW>(tbl, sm ? step : bal, os)) { missing++; give_up = true; break; }
		}
		if (give_up) continue;
		// first base of `at`: bits 2(K-1)..2(K-1)+1 of the NW-word value
		const int tb = 2 * (K - 1);
		uint32_t ch = 0;
#pragma unroll
		for (int w = 0; w < NW; w++)
			if (w == NW - 1 - (tb >> 6)) ch = (uint32_t)(at.w[w] >> (tb & 63)) & 3u;
		const uint32_t inf = ch | ((uint32_t)sm << 2) | (thin_stop << 3);
		if (end_out) {
			end_out[me] = idx[os];
			info_out[me] = (uint8_t)inf;
		} else {                                             // compact: only the nodes that have a walk, in any order
			const unsigned long long r = atomicAdd(cursor, 1ULL);
			if (r < max_rec) {
				rec[2 * r] = me | ((uint64_t)inf << 56);
				rec[2 * r + 1] = idx[os];
			}
		}
	}
	if (missing)
		atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}


// ---------------------------------------------------------------------------------------------------------------
// removeMinorOut's read-only part (cutTipPreGraph.c:1012-1076, clipKmerFromNode :591-1010, getmaxofprev/next :439-589)
// on the device mirror: for every junction (in > 1 or out > 1, not linear, not deleted) look its neighbours up,
// take the largest occurrence count per branching side and flag the neighbours whose count / max is under the
// threshold.  Links only disappear during the pass, so the cuts the ordered commit really makes are a subset of
// these; the host needs, per flagged junction and per flagged neighbour, who their neighbours ARE -- which never
// changes -- to run that commit without a single hash look-up.
// Record = 9 words: host index of the node, then (neighbour host index << 1 | smaller) or ~0 for LEFT 0..3, RIGHT 0..3.
// ---------------------------------------------------------------------------------------------------------------
template <int NW> __device__ inline Key<NW> key_prev_base(const Key<NW> &k, uint32_t b, int K)
{
	Key<NW> r = key_shr<NW>(k, 2);
	const int tb = 2 * (K - 1);
#pragma unroll
	for (int w = 0; w < NW; w++)
		if (w == NW - 1 - (tb >> 6)) r.w[w] |= (uint64_t)b << (tb & 63);
	return r;
}

template <int NW>
__device__ inline void neighbours_of(const Table<NW> &tbl, const uint64_t *__restrict__ idx, const Entry<NW> &e, int K, const Key<NW> &mask,
                                     uint64_t out[8], uint32_t cnt[8], uint32_t &missing)
{
	Key<NW> me;
#pragma unroll
	for (int w = 0; w < NW; w++) me.w[w] = e.key[w];
#pragma unroll
	for (int side = 0; side < 2; side++)
#pragma unroll
		for (uint32_t b = 0; b < 4; b++) {
			const int q = side * 4 + (int)b;
			out[q] = ~0ULL;
			cnt[q] = 0;
			if (!((e.val >> (24 * side + 6 * b)) & 63u)) continue;
			const Key<NW> word = side == 0 ? key_prev_base<NW>(me, b, K) : key_next_masked<NW>(me, b, mask);
			const Key<NW> bal = key_revcomp<NW>(word, K);
			const bool sm = !key_less<NW>(bal, word);
			uint64_t ns;
			if (!find_slot<NW>(tbl, sm ? word : bal, ns)) { missing++; continue; }
			out[q] = (idx[ns] << 1) | (uint64_t)sm;
			cnt[q] = (uint32_t)(tbl.ent[ns].val >> 48) | ((tbl.aux[ns] & 0xFFFFu) << 16);
		}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_minor_out_junctions(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, double threshold,
                                                             uint8_t *__restrict__ need, uint8_t *__restrict__ flagged,
                                                             uint64_t *__restrict__ rec, unsigned long long max_rec, unsigned long long *cursor,
                                                             Stats *stats)
{
	const uint64_t slots = tbl.mask + 1;
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	uint32_t missing = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		if (tbl.aux[s] & (AUX_LINEAR | AUX_DELETED)) continue;
		const uint32_t in = dev_degree(e.val & 0xFFFFFFu), out = dev_degree((e.val >> 24) & 0xFFFFFFu);
		if (in <= 1 && out <= 1) continue;
		uint64_t nb[8];
		uint32_t cnt[8];
		neighbours_of<NW>(tbl, idx, e, K, mask, nb, cnt, missing);
		bool any = false;
#pragma unroll
		for (int side = 0; side < 2; side++) {
			if ((side == 0 ? in : out) <= 1) continue;
			int best = 0;
#pragma unroll
			for (int b = 0; b < 4; b++)
				if (nb[side * 4 + b] != ~0ULL && (int)cnt[side * 4 + b] > best) best = (int)cnt[side * 4 + b];
			if (!best) continue;
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const int c = (int)cnt[side * 4 + b];
				if (nb[side * 4 + b] != ~0ULL && c && (double)c / best < threshold) {
					need[nb[side * 4 + b] >> 1] = 1;
					any = true;
				}
			}
		}
		if (!any) continue;
		const uint64_t me = idx[s];
		flagged[me] = 1;
		const unsigned long long r = atomicAdd(cursor, 1ULL);
		if (r < max_rec) {
			rec[r * 9] = me;
#pragma unroll
			for (int q = 0; q < 8; q++) rec[r * 9 + 1 + q] = nb[q];
		}
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}

// neighbours of the flagged neighbours (isolate() walks them), unless the node already has a junction record
template <int NW>
__global__ __launch_bounds__(TPB) void k_minor_out_candidates(Table<NW> tbl, const uint64_t *__restrict__ idx, int K,
                                                              const uint8_t *__restrict__ need, const uint8_t *__restrict__ flagged,
                                                              uint64_t *__restrict__ rec, unsigned long long max_rec, unsigned long long *cursor,
                                                              Stats *stats)
{
	const uint64_t slots = tbl.mask + 1;
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	uint32_t missing = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		const uint64_t me = idx[s];
		if (!need[me] || flagged[me]) continue;
		uint64_t nb[8];
		uint32_t cnt[8];
		neighbours_of<NW>(tbl, idx, e, K, mask, nb, cnt, missing);
		const unsigned long long r = atomicAdd(cursor, 1ULL);
		if (r < max_rec) {
			rec[r * 9] = me;
#pragma unroll
			for (int q = 0; q < 8; q++) rec[r * 9 + 1 + q] = nb[q];
		}
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}


// the host's own look-up index over its node array (csrc/host/graph/graph.c: open addressing on mix_key of the
// 4-word k-mer, 32-bit value = node index + 1), built here because the device already knows every node's index
__device__ inline uint64_t host_mix_key4(const uint64_t w[4])
{
	uint64_t h = 0x9E3779B97F4A7C15ULL;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		h ^= w[i];
		h ^= h >> 32; h *= 0xD6E8FEB86659FD93ULL; h ^= h >> 32;
	}
	return h;
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_build_host_index(Table<NW> tbl, const uint64_t *__restrict__ idx, unsigned int *__restrict__ index,
                                                          uint64_t index_mask)
{
	const uint64_t slots = tbl.mask + 1;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY) continue;
		uint64_t w[4] = {0, 0, 0, 0};
#pragma unroll
		for (int i = 0; i < NW; i++) w[4 - NW + i] = tbl.ent[s].key[i];
		uint64_t h = host_mix_key4(w) & index_mask;
		const unsigned int v = (unsigned int)(idx[s] + 1);
		while (atomicCAS(&index[h], 0u, v) != 0u) h = (h + 1) & index_mask;
	}
}


// k_set_paths when the device already knows each node's host index (sdt_gpu_set_node_index): no keys, no probing
template <int NW>
__global__ __launch_bounds__(TPB) void k_set_paths_by_index(Table<NW> tbl, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ info, uint64_t n,
                                                            Stats *stats)
{
	const uint64_t slots = tbl.mask + 1;
	uint32_t failed = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY) continue;
		const uint64_t i = idx[s];
		if (i < n) tbl.ent[s].val = info[i];
		else failed++;
	}
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}


// ---------------------------------------------------------------------------------------------------------------
// kmer2edges' read-only part (node2edge.c:46-191, stringBeads): from every node that is neither linear nor deleted,
// over each of its 8 ports (right links 0..3 on the stored strand, then left links 0..3 on the reverse strand) follow
// the chain of linear nodes to the first non-linear node.  Per port: the host index of that node, the port the chain
// arrives through, the chain length and whether the chain is its own reverse complement (bal_edge = 0,
// check_iden_kmerList :563-588).  The walk is forced after its first step, so the k-mer list equals its own
// reversed complement list exactly when the last k-mer is the complement of the first AND the second-to-last is the
// complement of the second -- four k-mers instead of the list.
// Record = 17 words: node index, then per port (far node index or ~0, length | far_port << 32 | bal_edge << 40).
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(TPB) void k_edge_ports(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, uint64_t max_steps,
                                                    uint64_t *__restrict__ rec, unsigned long long max_rec, unsigned long long *cursor, Stats *stats)
{
	const uint64_t slots = tbl.mask + 1;
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	const int tb = 2 * (K - 1);
	uint32_t missing = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		if (tbl.aux[s] & (AUX_LINEAR | AUX_DELETED)) continue;
		const unsigned long long r = atomicAdd(cursor, 1ULL);
		const bool keep = r < max_rec;
		if (keep) rec[r * 17] = idx[s];
		Key<NW> me;
#pragma unroll
		for (int w = 0; w < NW; w++) me.w[w] = e.key[w];
		const Key<NW> me_rc = key_revcomp<NW>(me, K);
		for (int p = 0; p < 8; p++) {
			uint64_t far = ~0ULL, meta = 0;
			const bool live = p < 4 ? ((e.val >> (24 + 6 * p)) & 63u) != 0 : ((e.val >> (6 * (p - 4))) & 63u) != 0;
			if (live) {
				const Key<NW> k0 = p < 4 ? me : me_rc;
				uint32_t b = p < 4 ? (uint32_t)p : ((uint32_t)(p - 4) ^ 2u);
				Key<NW> prev = k0, word = key_next_masked<NW>(k0, b, mask), k1 = word;
				uint64_t len = 1;
				bool ok = true, sm;
				uint64_t os;
				for (;;) {
					const Key<NW> bal = key_revcomp<NW>(word, K);
					sm = !key_less<NW>(bal, word);
					if (!find_slot<NW>(tbl, sm ? word : bal, os)) { missing++; ok = false; break; }
					if (!(tbl.aux[os] & AUX_LINEAR)) break;
					if (++len > max_steps) { missing++; ok = false; break; }
					const uint64_t ov = tbl.ent[os].val;
					b = sm ? first_link((ov >> 24) & 0xFFFFFFu) : (first_link(ov & 0xFFFFFFu) ^ 2u);
					prev = word;
					word = key_next_masked<NW>(word, b, mask);
				}
				if (ok) {
					// word = last k-mer, prev = second to last (== k0 when len == 1), k1 = second (== word when len == 1)
					uint32_t fc = 0;
#pragma unroll
					for (int w = 0; w < NW; w++)
						if (w == NW - 1 - (tb >> 6)) fc = (uint32_t)(prev.w[w] >> (tb & 63)) & 3u;
					const uint32_t far_port = sm ? 4u + fc : (fc ^ 2u);
					const Key<NW> rc0 = key_revcomp<NW>(k0, K), rc1 = key_revcomp<NW>(k1, K);
					const bool palin = key_eq<NW>(word, rc0) && key_eq<NW>(prev, rc1);
					far = idx[os];
					meta = len | ((uint64_t)far_port << 32) | ((uint64_t)(palin ? 0 : 1) << 40);
				}
			}
			if (keep) { rec[r * 17 + 1 + 2 * p] = far; rec[r * 17 + 2 + 2 * p] = meta; }
		}
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}
