// sdt_graph_kernels.cuh -- the graph phases of pregraph on the device mirror of the graph (cutTipPreGraph.c, node2edge.c):
// layout (visiting order of the reference's tables), read-only dry runs of the cutting passes, components of the commits
// (union-find), port walks of kmer2edges.  Included by sdt_gpu_graph.hip .
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// Graph-cleaning dry runs on the device (cutTipPreGraph.c).  The host owns the ORDER (layout replay, ordered commit
// of the few visits that write); what it needs from a sweep is the read-only part -- the walks -- and those are
// table look-ups, which this chip does at tens of G/s.  The device table mirrors the host graph: the host sends
// back the nodes it wrote (k_update_nodes) -- the nodes its own
// Mark1in1outNode marked included -- and the index each node has in its visiting order (k_set_index).
// ---------------------------------------------------------------------------------------------------------------
template <int NW> __device__ inline bool find_slot(const Table<NW> &tbl, const Key<NW> &k, uint64_t &slot_out)
{
	return table_find<NW>(tbl, k, slot_out);         // (either layout of the node table: sdt_table.cuh)
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
// the walk of one dead end (the node of entry e, aux word a): false = no walk from here (not a dead end, not `single` in the thin
// pass, longer than cut_len, or a link that leaves the graph: `missing`); else the slot of the end node and the info bits
// (first base of the last step | smaller << 2 | thin_stop << 3)
template <int NW>
__device__ inline bool tip_walk_from(const Table<NW> &tbl, const Entry<NW> &e, uint32_t a, int K, int thin, int cut_len, const Key<NW> &mask,
                                     uint32_t &missing, uint64_t &end_slot, uint32_t &inf)
{
	if (a & (AUX_LINEAR | AUX_DELETED)) return false;
	const bool single = (e.val >> 48) == 1 && (a & 0xFFFFu) == 0;
	if (thin && !single) return false;
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
		return false;
	}
	int steps = 1;
	uint32_t thin_stop = 0;
	Key<NW> step = key_next_masked<NW>(at, b, mask);
	Key<NW> bal = key_revcomp<NW>(step, K);
	bool sm = !key_less<NW>(bal, step);               // KmerLarger(word, bal) -> take bal, smaller = 0
	uint64_t os;
	if (!find_slot<NW>(tbl, sm ? step : bal, os)) { missing++; return false; }
	for (;;) {
		const uint32_t oa = tbl.aux[os];
		if (!(oa & AUX_LINEAR)) break;
		steps++;
		const uint64_t ov = tbl.ent[os].val;
		if (thin && !((ov >> 48) == 1 && (oa & 0xFFFFu) == 0)) { thin_stop = 1; break; }
		if (steps > cut_len) return false;
		at = step;
		b = sm ? first_link((ov >> 24) & 0xFFFFFFu) : (first_link(ov & 0xFFFFFFu) ^ 2u);
		step = key_next_masked<NW>(at, b, mask);
		bal = key_revcomp<NW>(step, K);
		sm = !key_less<NW>(bal, step);
		if (!find_slot<NW>(tbl, sm ? step : bal, os)) { missing++; return false; }
	}
	// first base of `at`: bits 2(K-1)..2(K-1)+1 of the NW-word value
	const int tb = 2 * (K - 1);
	uint32_t ch = 0;
#pragma unroll
	for (int w = 0; w < NW; w++)
		if (w == NW - 1 - (tb >> 6)) ch = (uint32_t)(at.w[w] >> (tb & 63)) & 3u;
	inf = ch | ((uint32_t)sm << 2) | (thin_stop << 3);
	end_slot = os;
	return true;
}

template <int NW> __device__ inline Key<NW> key_mask_of(int K)
{
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	return mask;
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_tip_walks(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, int thin, int cut_len,
                                                   uint64_t *__restrict__ end_out, uint8_t *__restrict__ info_out, Stats *stats,
                                                   uint64_t *__restrict__ rec = nullptr, unsigned long long max_rec = 0,
                                                   unsigned long long *cursor = nullptr, int rec_stride = 2)
{
	const uint64_t slots = tbl.slots();
	const Key<NW> mask = key_mask_of<NW>(K);
	uint32_t missing = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		const uint64_t me = idx[s];
		if (end_out) {
			end_out[me] = ~0ULL;
			info_out[me] = 0;
		}
		uint64_t os;
		uint32_t inf;
		if (!tip_walk_from<NW>(tbl, e, tbl.aux[s], K, thin, cut_len, mask, missing, os, inf)) continue;
		if (end_out) {
			end_out[me] = idx[os];
			info_out[me] = (uint8_t)inf;
		} else {                                             // compact: only the nodes that have a walk, in any order
			const unsigned long long r = atomicAdd(cursor, 1ULL);
			if (r < max_rec) {
				rec[rec_stride * r] = me | ((uint64_t)inf << 56);
				rec[rec_stride * r + 1] = idx[os];
			}
		}
	}
	if (missing)
		atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}

// The same in two steps, for the labelled dry runs.  One lane in fifty of the scan has a walk, and a walk is a chain of up to 2K
// dependent look-ups: a wave of the scan waits for its one or two walkers with all other lanes idle (430 ms for the two tip passes
// at 678 M nodes).  k_tip_starts lists the dead ends (their slots, in chunks: sdt_append.cuh), k_tip_walks_list gives every lane of a
// wave a walk of its own.
template <int NW>
__global__ __launch_bounds__(TPB) void k_tip_starts(Table<NW> tbl, int thin, unsigned long long *__restrict__ list, ApOut ap)
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const uint64_t slots = tbl.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		const uint32_t a = tbl.aux[s];
		if (a & (AUX_LINEAR | AUX_DELETED)) continue;
		if (thin && !((e.val >> 48) == 1 && (a & 0xFFFFu) == 0)) continue;
		const uint32_t in = dev_degree(e.val & 0xFFFFFFu), out = dev_degree((e.val >> 24) & 0xFFFFFFu);
		if (!((in == 0 && out == 1) || (in == 1 && out == 0))) continue;
		const unsigned long long r = ap_append(s_app, ap);
		if (r != AP_NONE) list[r] = s;
	}
	ap_finish_mark(s_app, ap);
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_tip_walks_list(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, int thin, int cut_len,
                                                        const unsigned long long *__restrict__ list, unsigned long long n_list, Stats *stats,
                                                        uint64_t *__restrict__ rec, int rec_stride, ApOut ap)
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const Key<NW> mask = key_mask_of<NW>(K);
	uint32_t missing = 0;
	for (unsigned long long k = blockIdx.x * (unsigned long long)TPB + threadIdx.x; k < n_list; k += (unsigned long long)gridDim.x * TPB) {
		const unsigned long long s = list[k];
		if (s == AP_NONE) continue;
		const Entry<NW> e = tbl.ent[s];
		uint64_t os;
		uint32_t inf;
		if (!tip_walk_from<NW>(tbl, e, tbl.aux[s], K, thin, cut_len, mask, missing, os, inf)) continue;
		const unsigned long long r = ap_append(s_app, ap);
		if (r != AP_NONE) {
			rec[rec_stride * r] = idx[s] | ((uint64_t)inf << 56);
			rec[rec_stride * r + 1] = idx[os];
		}
	}
	ap_finish(s_app, ap);
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
                                                             Stats *stats, int rec_stride = 9, ApOut ap = ApOut{nullptr, 0, nullptr, nullptr})
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const uint64_t slots = tbl.slots();
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
		unsigned long long r;
		if (ap.cursor) r = ap_append(s_app, ap);
		else { r = atomicAdd(cursor, 1ULL); if (r >= max_rec) r = AP_NONE; }
		if (r != AP_NONE) {
			rec[r * rec_stride] = me;
#pragma unroll
			for (int q = 0; q < 8; q++) rec[r * rec_stride + 1 + q] = nb[q];
			if (rec_stride >= 13) {                              // the neighbours' occurrence counts, two per word
#pragma unroll
				for (int q = 0; q < 8; q += 2) rec[r * rec_stride + 9 + q / 2] = (uint64_t)cnt[q] | ((uint64_t)cnt[q + 1] << 32);
			}
		}
	}
	if (ap.cursor) ap_finish(s_app, ap);
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}

// neighbours of the flagged neighbours (isolate() walks them), unless the node already has a junction record
template <int NW>
__global__ __launch_bounds__(TPB) void k_minor_out_candidates(Table<NW> tbl, const uint64_t *__restrict__ idx, int K,
                                                              const uint8_t *__restrict__ need, const uint8_t *__restrict__ flagged,
                                                              uint64_t *__restrict__ rec, unsigned long long max_rec, unsigned long long *cursor,
                                                              Stats *stats, int rec_stride = 9, ApOut ap = ApOut{nullptr, 0, nullptr, nullptr})
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const uint64_t slots = tbl.slots();
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
		unsigned long long r;
		if (ap.cursor) r = ap_append(s_app, ap);
		else { r = atomicAdd(cursor, 1ULL); if (r >= max_rec) r = AP_NONE; }
		if (r != AP_NONE) {
			rec[r * rec_stride] = me;
#pragma unroll
			for (int q = 0; q < 8; q++) rec[r * rec_stride + 1 + q] = nb[q];
			if (rec_stride >= 13) {                              // the neighbours' occurrence counts, two per word
#pragma unroll
				for (int q = 0; q < 8; q += 2) rec[r * rec_stride + 9 + q / 2] = (uint64_t)cnt[q] | ((uint64_t)cnt[q + 1] << 32);
			}
		}
	}
	if (ap.cursor) ap_finish(s_app, ap);
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
	const uint64_t slots = tbl.slots();
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
	const uint64_t slots = tbl.slots();
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


// ===============================================================================================================
// Layout: the reference's visiting order (set 0..p-1, slot 0..size-1 of its per-thread tables)
// ---------------------------------------------------------------------------------------------------------------
// A node's set is hash_kmer(key) % p (hashFunction.c:83-122: table-driven CRC-32 over the bytes of the variant's Kmer
// struct with a SIGNED state, low 24 bits), its place inside the set follows from the order in which the set's distinct
// keys first occurred (put_kmerset / encap_kmerset, newhash.c:293-462).  The device knows both: it sorts the nodes by
// (set, first-occurrence ordinal) and hands the host the KEYS in that order; the host replays only the probing of each
// set (graph.c) and answers with order[v] = rank of the node at visiting position v; k_layout_apply then numbers the
// nodes (idx[slot] = v) and lays the node arrays out in visiting order for the export.
// ===============================================================================================================
__device__ inline void crc_table_to_lds(int32_t *s_crc)
{
	for (uint32_t n = threadIdx.x; n < 256; n += blockDim.x) {
		uint32_t c = n;
#pragma unroll
		for (int b = 0; b < 8; b++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
		s_crc[n] = (int32_t)c;
	}
	__syncthreads();
}

// sort key = set << 56 | first-occurrence ordinal (< 2^56), value = table slot
template <int NW>
__global__ __launch_bounds__(TPB) void k_layout_keys(Table<NW> tbl, uint32_t p, int nw_variant, uint64_t *__restrict__ skey,
                                                     uint64_t *__restrict__ sval, unsigned long long max_nodes, unsigned long long *cursor, Stats *stats)
{
	__shared__ int32_t s_crc[256];
	crc_table_to_lds(s_crc);
	__shared__ unsigned long long s_res[1 + TPB / 64];
	const uint64_t slots = tbl.slots();
	uint32_t bad = 0;
	// eight slots per lane and ONE reservation per workgroup for all of them (the output order is free: the pairs are sorted next)
	constexpr int IT = 8;
	for (uint64_t base = blockIdx.x * (uint64_t)TPB * IT; base < slots; base += (uint64_t)gridDim.x * TPB * IT) {
		uint32_t occ = 0;
#pragma unroll
		for (int j = 0; j < IT; j++) {
			const uint64_t s = base + (uint64_t)j * TPB + threadIdx.x;
			if (s < slots && tbl.ent[s].key[0] != KEY_EMPTY) occ |= 1u << j;
		}
		unsigned long long pos = ap_block_reserve(__popc(occ), cursor, s_res);
		for (int j = 0; j < IT; j++) {
			if (!((occ >> j) & 1u)) continue;
			const uint64_t s = base + (uint64_t)j * TPB + threadIdx.x;
			const Entry<NW> e = tbl.ent[s];
			int32_t crc = ~0;
			// raw bytes of the variant's Kmer struct: words most significant first, each little endian; a variant wider than the
			// device key has zero words in front
			for (int w = 0; w < nw_variant; w++) {
				const int kw = w - (nw_variant - NW);
				const uint64_t word = kw >= 0 ? e.key[kw] : 0ULL;
#pragma unroll
				for (int b = 0; b < 8; b++)
					crc = s_crc[(crc ^ (int32_t)((word >> (8 * b)) & 0xFF)) & 0xFF] ^ (crc >> 8);     // >> on a negative int shifts sign bits in
			}
			crc = ~crc;
			const uint64_t set = ((uint64_t)(int64_t)crc & 0xFFFFFFULL) % p;
			const uint64_t first = tbl.first[s];
			if (first >> 56) bad++;
			const unsigned long long at = pos++;
			if (at >= max_nodes) { bad++; continue; }
			skey[at] = (set << 56) | (first & 0x00FFFFFFFFFFFFFFULL);
			sval[at] = s;
		}
	}
	if (bad) atomicAdd(&stats->probe_fail, (unsigned long long)bad);
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_layout_gather_keys(Table<NW> tbl, const uint64_t *__restrict__ sval, uint64_t n, uint64_t *__restrict__ keys)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> *e = tbl.ent + sval[i];
#pragma unroll
		for (int w = 0; w < NW; w++) keys[i * NW + w] = e->key[w];
	}
}

// set_start[s] = first rank whose set is >= s (s = 0..p)
__global__ void k_layout_set_starts(const uint64_t *__restrict__ skey, uint64_t n, uint32_t p, uint64_t *__restrict__ set_start)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s > p) return;
	uint64_t lo = 0, hi = n;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if ((skey[mid] >> 56) < s) lo = mid + 1; else hi = mid;
	}
	set_start[s] = lo;
}

// order[v] = rank of the node at visiting position v: number the nodes, remember their slots
__global__ __launch_bounds__(TPB) void k_layout_apply(const uint64_t *__restrict__ sval, const uint64_t *__restrict__ order, uint64_t n,
                                                      uint64_t *__restrict__ idx, uint64_t *__restrict__ slot_of, Stats *stats)
{
	uint32_t bad = 0;
	for (uint64_t v = blockIdx.x * (uint64_t)TPB + threadIdx.x; v < n; v += (uint64_t)gridDim.x * TPB) {
		const uint64_t r = order[v];
		if (r >= n) { bad++; continue; }
		const uint64_t s = sval[r];
		idx[s] = v;
		slot_of[v] = s;
	}
	if (bad) atomicAdd(&stats->probe_fail, (unsigned long long)bad);
}

// the nodes in visiting order, kmer_t-shaped (as k_export)
template <int NW>
__global__ __launch_bounds__(TPB) void k_export_ordered(Table<NW> tbl, const uint64_t *__restrict__ slot_of, uint64_t v0, uint64_t n,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ l_links,
                                                        uint32_t *__restrict__ r_flags, uint32_t *__restrict__ count)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		const uint64_t s = slot_of[v0 + i];
		const Entry<NW> e = tbl.ent[s];
		const uint32_t aux = tbl.aux[s];
		const uint32_t cnt = ((aux & 0xFFFFu) << 16) | (uint32_t)(e.val >> 48);
		if (keys) {
#pragma unroll
			for (int w = 0; w < NW; w++) keys[i * NW + w] = e.key[w];
		}
		if (l_links) l_links[i] = (uint32_t)(e.val & 0xFFFFFFu);
		if (r_flags)
			r_flags[i] = (uint32_t)((e.val >> 24) & 0xFFFFFFu) | ((aux & AUX_LINEAR) ? 1u << 24 : 0u) |
			             ((aux & AUX_DELETED) ? 1u << 25 : 0u) | (cnt == 1 ? 1u << 27 : 0u);
		if (count) count[i] = cnt;
	}
}

// links + flags of the nodes the host wrote, addressed by node index (no keys cross the link, no look-ups)
template <int NW>
__global__ __launch_bounds__(TPB) void k_update_by_index(Table<NW> tbl, const uint64_t *__restrict__ slot_of, uint64_t n_nodes,
                                                         const uint64_t *__restrict__ node, const uint32_t *__restrict__ l_links,
                                                         const uint32_t *__restrict__ r_flags, uint64_t n, Stats *stats)
{
	uint32_t failed = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		if (node[i] >= n_nodes) { failed++; continue; }
		const uint64_t slot = slot_of[node[i]];
		const uint64_t v = tbl.ent[slot].val;
		tbl.ent[slot].val = (v & 0xFFFF000000000000ULL) | ((uint64_t)(r_flags[i] & 0xFFFFFFu) << 24) | (uint64_t)(l_links[i] & 0xFFFFFFu);
		const uint32_t a = tbl.aux[slot] & 0xFFFFu;
		tbl.aux[slot] = a | ((r_flags[i] >> 24 & 1u) ? AUX_LINEAR : 0u) | ((r_flags[i] >> 25 & 1u) ? AUX_DELETED : 0u);
	}
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}


// ===============================================================================================================
// removeMinorOut's COMMIT on the device (cutTipPreGraph.c:591-1010; csrc/host/graph/cuttip.c visit_minor_out / prune_side /
// isolate are the host's form).  The labelled dry run left, on the device: the junction records sorted by (component, node) --
// node, its 8 neighbours as (index << 1 | orientation), their occurrence counts -- and the records of the neighbours a visit
// may cut.  A visit reads and writes only nodes of its own component, so components commute; inside one the visits run in
// record order = the reference's visiting order.  One lane per component.
// ===============================================================================================================
__global__ __launch_bounds__(TPB) void k_mo_comp_flags(const uint64_t *__restrict__ rec, uint64_t nj, int stride, uint32_t *__restrict__ flag)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r <= nj; r += (uint64_t)gridDim.x * TPB)
		flag[r] = r < nj && (r == 0 || rec[r * stride + stride - 1] != rec[(r - 1) * stride + stride - 1]);
}

// cstart[c] = first record of component c (rank = exclusive scan of the flags; cstart[ncomp] = nj is written by the host side);
// largest[0] = the largest component
__global__ __launch_bounds__(TPB) void k_mo_comp_starts(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ rank, uint64_t nj,
                                                        uint32_t *__restrict__ cstart)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r < nj; r += (uint64_t)gridDim.x * TPB)
		if (flag[r]) cstart[rank[r]] = (uint32_t)r;
}

__global__ __launch_bounds__(TPB) void k_mo_comp_largest(const uint32_t *__restrict__ cstart, uint64_t ncomp, unsigned long long *largest)
{
	unsigned long long m = 0;
	for (uint64_t c = blockIdx.x * (uint64_t)TPB + threadIdx.x; c < ncomp; c += (uint64_t)gridDim.x * TPB) {
		const unsigned long long sz = cstart[c + 1] - cstart[c];
		if (sz > m) m = sz;
	}
	if (m) atomicMax(largest, m);
}

// recidx[node] = 1 + the record that lists the node's neighbours (junction records and the records of the neighbours to cut)
__global__ __launch_bounds__(TPB) void k_mo_recidx(const uint64_t *__restrict__ rec, uint64_t nr, int stride, uint32_t *__restrict__ recidx)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r < nr; r += (uint64_t)gridDim.x * TPB)
		recidx[rec[r * stride]] = (uint32_t)r + 1u;
}

// a lane re-reads what it (and only it, during this kernel) wrote: real loads and stores every time, but no cache-bypassing scope --
// they may stop in the CU's L1 and the XCD's L2, which are coherent for one lane's own accesses (volatile accesses went to the
// memory side: 25 us per visit, 1.4 s for the 56 288 visits of the largest component of the 200 M-read job)
template <class T> __device__ inline T mo_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
template <class T> __device__ inline void mo_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

template <int NW> __device__ inline uint32_t mo_first_base(const Entry<NW> *e, int K)
{
	const int bit = 2 * (K - 1), wi = NW - 1 - bit / 64;
	return (uint32_t)(e->key[wi] >> (bit % 64)) & 3u;
}

// isolate(q) (cuttip.c): q is deleted, every neighbour forgets its link to q and has its `linear` re-derived
template <int NW>
__device__ inline void mo_isolate(const Table<NW> &tbl, const uint64_t *__restrict__ slot_of, int K, const uint64_t *__restrict__ rec, int stride,
                                  const uint32_t *__restrict__ recidx, uint8_t *__restrict__ dirty, uint64_t q, uint32_t &errors)
{
	const uint64_t sq = slot_of[q];
	Entry<NW> *eq = tbl.ent + sq;
	const uint32_t ch_last = (uint32_t)eq->key[NW - 1] & 3u, ch_first = mo_first_base<NW>(eq, K);
	mo_st(&tbl.aux[sq], mo_ld(&tbl.aux[sq]) | AUX_DELETED);
	dirty[q] = 1;
	const uint32_t rq = recidx[q];
	if (!rq) { errors++; return; }
	const uint64_t *Q = rec + (uint64_t)(rq - 1u) * stride;
#pragma unroll 1
	for (int side = 0; side < 2; side++)
#pragma unroll 1
		for (int b = 0; b < 4; b++) {
			// (q's own links are read again every time: a neighbour of q can be q itself)
			const uint64_t vq = mo_ld(&eq->val);
			if (!(((side == 0 ? vq : vq >> 24) >> (6 * b)) & 63u)) continue;
			const uint64_t nb = Q[1 + side * 4 + b];
			if (nb == ~0ULL) { errors++; continue; }
			const uint64_t x = nb >> 1, sx = slot_of[x];
			const bool sm = (nb & 1u) != 0;
			// unlink_next(x, last base of q, sm) for q's left neighbours, unlink_prev(y, first base of q, sm) for its right ones
			int drop;                                    // field of val to clear: 0..3 left links, 4..7 right links
			if (side == 0) drop = sm ? 4 + (int)ch_last : (int)(ch_last ^ 2u);
			else drop = sm ? (int)ch_first : 4 + (int)(ch_first ^ 2u);
			uint64_t vx = mo_ld(&tbl.ent[sx].val);
			vx &= ~(63ULL << (6 * drop));
			mo_st(&tbl.ent[sx].val, vx);
			const bool lin = dev_degree(vx & 0xFFFFFFu) == 1 && dev_degree((vx >> 24) & 0xFFFFFFu) == 1;
			const uint32_t ax = mo_ld(&tbl.aux[sx]);
			mo_st(&tbl.aux[sx], lin ? (ax | AUX_LINEAR) : (ax & ~AUX_LINEAR));
			dirty[x] = 1;
		}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_mo_commit(Table<NW> tbl, const uint64_t *__restrict__ slot_of, int K, double threshold,
                                                   const uint64_t *__restrict__ rec, int stride, const uint32_t *__restrict__ cstart, uint64_t ncomp,
                                                   const uint32_t *__restrict__ recidx, uint8_t *__restrict__ dirty, unsigned long long *counters,
                                                   uint64_t max_component)
{
	unsigned long long off = 0;
	uint32_t errors = 0;
	for (uint64_t c = blockIdx.x * (uint64_t)TPB + threadIdx.x; c < ncomp; c += (uint64_t)gridDim.x * TPB) {
		const uint32_t r1 = cstart[c + 1];
		if ((uint64_t)(r1 - cstart[c]) > max_component) continue;      // (left to the host's threads: k_mo_skipped_*)
#pragma unroll 1
		for (uint32_t r = cstart[c]; r < r1; r++) {
			const uint64_t *R = rec + (uint64_t)r * stride;
			const uint64_t sn = slot_of[R[0]];
			const uint64_t *pv = &tbl.ent[sn].val;
			if (mo_ld(&tbl.aux[sn]) & (AUX_LINEAR | AUX_DELETED)) continue;
			const uint64_t v0 = mo_ld(pv);
			const uint32_t in = dev_degree(v0 & 0xFFFFFFu), out = dev_degree((v0 >> 24) & 0xFFFFFFu);      // both sampled before any cut (:616-617)
			if (in <= 1 && out <= 1) continue;
#pragma unroll 1
			for (int side = 0; side < 2; side++) {
				if ((side == 0 ? in : out) <= 1) continue;
				int best = 0;
				const uint64_t vs = mo_ld(pv);
				for (int b = 0; b < 4; b++)
					if (((side == 0 ? vs : vs >> 24) >> (6 * b)) & 63u) {
						const uint64_t cw = R[9 + (side * 4 + b) / 2];
						const int cnt = (int)(uint32_t)((side * 4 + b) & 1 ? cw >> 32 : cw);
						if (cnt > best) best = cnt;
					}
				if (!best) continue;
#pragma unroll 1
				for (int b = 0; b < 4; b++) {
					const uint64_t vl = mo_ld(pv);                 // live: an earlier cut may have removed the link
					if (!(((side == 0 ? vl : vl >> 24) >> (6 * b)) & 63u)) continue;
					const uint64_t cw = R[9 + (side * 4 + b) / 2];
					const int cnt = (int)(uint32_t)((side * 4 + b) & 1 ? cw >> 32 : cw);
					if (cnt && (double)cnt / best < threshold) {
						off++;
						const uint64_t nb = R[1 + side * 4 + b];
						if (nb == ~0ULL) { errors++; continue; }
						mo_isolate<NW>(tbl, slot_of, K, rec, stride, recidx, dirty, nb >> 1, errors);
					}
				}
			}
		}
	}
	if (off) atomicAdd(&counters[0], off);
	if (errors) atomicAdd(&counters[1], (unsigned long long)errors);
}

// the junction records of the components k_mo_commit left alone, in order: sel[r] = 1 for them (rank = exclusive scan of the start
// flags: the component of record r is rank[r] + flag[r] - 1), then gathered to the positions an exclusive scan of sel gives
__global__ __launch_bounds__(TPB) void k_mo_skipped_sel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ rank, const uint32_t *__restrict__ cstart,
                                                        uint64_t nj, uint64_t max_component, uint32_t *__restrict__ sel)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r <= nj; r += (uint64_t)gridDim.x * TPB) {
		uint32_t v = 0;
		if (r < nj) {
			const uint32_t c = rank[r] + flag[r] - 1u;
			v = (uint64_t)(cstart[c + 1] - cstart[c]) > max_component;
		}
		sel[r] = v;
	}
}

__global__ __launch_bounds__(TPB) void k_mo_skipped_gather(const uint64_t *__restrict__ rec, int stride, const uint32_t *__restrict__ sel,
                                                           const uint32_t *__restrict__ pos, uint64_t nj, uint64_t *__restrict__ out)
{
	for (uint64_t w = blockIdx.x * (uint64_t)TPB + threadIdx.x; w < nj * (uint64_t)stride; w += (uint64_t)gridDim.x * TPB) {
		const uint64_t r = w / (uint64_t)stride;
		if (sel[r]) out[(uint64_t)pos[r] * stride + w % (uint64_t)stride] = rec[w];
	}
}

// size_of[label] = visits of the component with that label
__global__ __launch_bounds__(TPB) void k_mo_label_sizes(const uint64_t *__restrict__ rec, int stride, const uint32_t *__restrict__ cstart, uint64_t ncomp,
                                                        uint32_t *__restrict__ size_of)
{
	for (uint64_t c = blockIdx.x * (uint64_t)TPB + threadIdx.x; c < ncomp; c += (uint64_t)gridDim.x * TPB)
		size_of[rec[(uint64_t)cstart[c] * stride + stride - 1]] = cstart[c + 1] - cstart[c];
}

// the neighbour records [nj, nr) that belong to components left to the host
__global__ __launch_bounds__(TPB) void k_mo_skipped_sel2(const uint64_t *__restrict__ rec, int stride, uint64_t nj, uint64_t nr, const uint32_t *__restrict__ size_of,
                                                         uint64_t max_component, uint32_t *__restrict__ sel)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r <= nr - nj; r += (uint64_t)gridDim.x * TPB)
		sel[r] = r < nr - nj && (uint64_t)size_of[rec[(nj + r) * stride + stride - 1]] > max_component;
}

// mark_linear over the nodes the commit wrote (cuttip.c: mark_linear_dirty) + their number
template <int NW>
__global__ __launch_bounds__(TPB) void k_mo_mark(Table<NW> tbl, const uint64_t *__restrict__ slot_of, uint64_t nn, const uint8_t *__restrict__ dirty,
                                                 unsigned long long *counters)
{
	unsigned long long marked = 0, written = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < nn; i += (uint64_t)gridDim.x * TPB) {
		if (!dirty[i]) continue;
		written++;
		const uint64_t s = slot_of[i];
		const uint32_t a = tbl.aux[s];
		if (a & (AUX_LINEAR | AUX_DELETED)) continue;
		const uint64_t v = tbl.ent[s].val;
		if (dev_degree(v & 0xFFFFFFu) == 1 && dev_degree((v >> 24) & 0xFFFFFFu) == 1) {
			tbl.aux[s] = a | AUX_LINEAR;
			marked++;
		}
	}
	if (marked) atomicAdd(&counters[2], marked);
	if (written) atomicAdd(&counters[3], written);
}

// the written nodes as (index, l_links, r_links | linear << 24 | deleted << 25): what sdt_gpu_update_nodes_by_index takes, the other way
template <int NW>
__global__ __launch_bounds__(TPB) void k_mo_emit(Table<NW> tbl, const uint64_t *__restrict__ slot_of, uint64_t nn, const uint8_t *__restrict__ dirty,
                                                 unsigned long long *cursor, uint64_t cap, uint64_t *__restrict__ node, uint32_t *__restrict__ l_links,
                                                 uint32_t *__restrict__ r_flags)
{
	// eight nodes per lane, one reservation per workgroup (one per wave and iteration, 10.6 M on one word, took 126 ms of this kernel)
	__shared__ unsigned long long s_res[1 + TPB / 64];
	constexpr int IT = 8;
	for (uint64_t base = blockIdx.x * (uint64_t)TPB * IT; base < nn; base += (uint64_t)gridDim.x * TPB * IT) {
		uint32_t occ = 0;
#pragma unroll
		for (int j = 0; j < IT; j++) {
			const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
			if (i < nn && dirty[i]) occ |= 1u << j;
		}
		unsigned long long at = ap_block_reserve(__popc(occ), cursor, s_res);
		for (int j = 0; j < IT; j++) {
			if (!((occ >> j) & 1u)) continue;
			const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
			const unsigned long long mine = at++;
			if (mine >= cap) continue;
			const uint64_t s = slot_of[i], v = tbl.ent[s].val;
			const uint32_t a = tbl.aux[s];
			node[mine] = i;
			l_links[mine] = (uint32_t)(v & 0xFFFFFFu);
			r_flags[mine] = (uint32_t)((v >> 24) & 0xFFFFFFu) | ((a & AUX_LINEAR) ? 1u << 24 : 0u) | ((a & AUX_DELETED) ? 1u << 25 : 0u);
		}
	}
}

// ===============================================================================================================
// Components of the ordered commits (csrc/host/graph/cuttip.c): lock-free union-find over node indices.
// parent[] only ever changes from "root" to "child of a smaller root" (and by path halving, to an ancestor), so a stale value
// still leads up the same tree; the per-XCD L2s are not coherent, so parent[] is read with device-scope atomic loads and
// written with device-scope CAS only -- those resolve at the memory side.
// ===============================================================================================================
__device__ inline uint32_t uf_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline uint32_t uf_find(uint32_t *parent, uint32_t x)
{
	for (;;) {
		const uint32_t p = uf_load(parent + x);
		if (p == x) return x;
		const uint32_t gp = uf_load(parent + p);
		if (gp != p) atomicCAS(parent + x, p, gp);          // path halving
		x = p;
	}
}

__device__ inline void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
	for (;;) {
		a = uf_find(parent, a);
		b = uf_find(parent, b);
		if (a == b) return;
		if (a < b) { const uint32_t t = a; a = b; b = t; }   // the larger root goes under the smaller
		if (atomicCAS(parent + a, a, b) == a) return;
	}
}

__global__ __launch_bounds__(TPB) void k_uf_init(uint32_t *__restrict__ parent, uint64_t n)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) parent[i] = (uint32_t)i;
}

// records of `stride` words: word 0 = node (low 56 bits); unite the node with the node indices found in words [w0, w1), each
// shifted right by `shift` (an entry of ~0 is no node)
__global__ __launch_bounds__(TPB) void k_uf_records(uint32_t *parent, const uint64_t *__restrict__ rec, uint64_t n, int stride, int w0, int w1, int shift)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r < n; r += (uint64_t)gridDim.x * TPB) {
		const uint32_t me = (uint32_t)(rec[r * stride] & 0x00FFFFFFFFFFFFFFULL);
		for (int w = w0; w < w1; w++) {
			const uint64_t x = rec[r * stride + w];
			if (x != ~0ULL) uf_union(parent, me, (uint32_t)(x >> shift));
		}
	}
}

// label word of every record = root of its node; sort key = label << 32 | node
__global__ __launch_bounds__(TPB) void k_uf_label(uint32_t *parent, uint64_t *__restrict__ rec, uint64_t n, int stride, int label_word,
                                                  uint64_t *__restrict__ skey, uint32_t *__restrict__ sval)
{
	for (uint64_t r = blockIdx.x * (uint64_t)TPB + threadIdx.x; r < n; r += (uint64_t)gridDim.x * TPB) {
		const uint32_t me = (uint32_t)(rec[r * stride] & 0x00FFFFFFFFFFFFFFULL);
		const uint32_t root = uf_find(parent, me);
		rec[r * stride + label_word] = root;
		skey[r] = ((uint64_t)root << 32) | me;
		sval[r] = (uint32_t)r;
	}
}

// label word of records that are not sorted (the neighbours to cut behind the junction records)
__global__ __launch_bounds__(TPB) void k_uf_label_only(uint32_t *parent, uint64_t *__restrict__ rec, uint64_t r0, uint64_t n, int stride, int label_word)
{
	for (uint64_t r = r0 + blockIdx.x * (uint64_t)TPB + threadIdx.x; r < n; r += (uint64_t)gridDim.x * TPB)
		rec[r * stride + label_word] = uf_find(parent, (uint32_t)(rec[r * stride] & 0x00FFFFFFFFFFFFFFULL));
}

__global__ __launch_bounds__(TPB) void k_gather_records(const uint64_t *__restrict__ rec, const uint32_t *__restrict__ perm, uint64_t n, int stride,
                                                        uint64_t *__restrict__ out)
{
	const uint64_t total = n * (uint64_t)stride;
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < total; i += (uint64_t)gridDim.x * TPB) {
		const uint64_t r = i / (uint64_t)stride, w = i - r * (uint64_t)stride;
		out[i] = rec[(uint64_t)perm[r] * stride + w];
	}
}

// removeMinorTips' components: every node that is neither linear nor deleted is united with the first non-linear node behind
// each of its live ports when at most max_linear linear nodes lie in between (the chains a walk of <= cut_len steps can cross)
// (in two steps, as the tip walks: k_port_starts lists slot << 3 | port for every live port of such a node, k_port_union_list
// gives every lane one port to walk)
template <int NW>
__global__ __launch_bounds__(TPB) void k_port_starts(Table<NW> tbl, unsigned long long *__restrict__ list, ApOut ap)
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const uint64_t slots = tbl.slots();
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		const Entry<NW> e = tbl.ent[s];
		if (e.key[0] == KEY_EMPTY) continue;
		if (tbl.aux[s] & (AUX_LINEAR | AUX_DELETED)) continue;
		for (int p = 0; p < 8; p++) {
			const bool live = p < 4 ? ((e.val >> (24 + 6 * p)) & 63u) != 0 : ((e.val >> (6 * (p - 4))) & 63u) != 0;
			if (!live) continue;
			const unsigned long long r = ap_append(s_app, ap);
			if (r != AP_NONE) list[r] = (s << 3) | (unsigned long long)p;
		}
	}
	ap_finish_mark(s_app, ap);
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_port_union_list(Table<NW> tbl, const uint64_t *__restrict__ idx, int K, int max_linear,
                                                         const unsigned long long *__restrict__ list, unsigned long long n_list, uint32_t *parent, Stats *stats)
{
	const Key<NW> mask = key_mask_of<NW>(K);
	uint32_t missing = 0;
	for (unsigned long long k = blockIdx.x * (unsigned long long)TPB + threadIdx.x; k < n_list; k += (unsigned long long)gridDim.x * TPB) {
		const unsigned long long le = list[k];
		if (le == AP_NONE) continue;
		const uint64_t s = le >> 3;
		const int p = (int)(le & 7u);
		const Entry<NW> e = tbl.ent[s];
		Key<NW> me;
#pragma unroll
		for (int w = 0; w < NW; w++) me.w[w] = e.key[w];
		if (p >= 4) me = key_revcomp<NW>(me, K);
		uint32_t b = p < 4 ? (uint32_t)p : ((uint32_t)(p - 4) ^ 2u);
		Key<NW> word = key_next_masked<NW>(me, b, mask);
		int passed = 0;
		bool ok = true;
		uint64_t os;
		for (;;) {
			const Key<NW> bal = key_revcomp<NW>(word, K);
			const bool sm = !key_less<NW>(bal, word);
			if (!find_slot<NW>(tbl, sm ? word : bal, os)) { missing++; ok = false; break; }
			if (!(tbl.aux[os] & AUX_LINEAR)) break;
			if (++passed > max_linear) { ok = false; break; }
			const uint64_t ov = tbl.ent[os].val;
			b = sm ? first_link((ov >> 24) & 0xFFFFFFu) : (first_link(ov & 0xFFFFFFu) ^ 2u);
			word = key_next_masked<NW>(word, b, mask);
		}
		if (ok) uf_union(parent, (uint32_t)idx[s], (uint32_t)idx[os]);
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}


// ===============================================================================================================
// kmer2edges on the device (node2edge.c:46-561): every chain of linear nodes between two nodes that are neither linear
// nor deleted is one edge.  The reference emits it from whichever end it visits first (set, slot order; right links 0..3
// on the stored strand, then left links 0..3 on the other), numbers the edges in that order (a chain that is not its own
// reverse complement takes two ids) and stamps the interior nodes with the id.  With the nodes numbered in visiting order
// all of that is data parallel: an edge belongs to the smaller of its two (node, port) ends, its id is a prefix sum.
//   k_edge_starts          flag the nodes that start edges; initial path word of every node (second read pass, sdt_map_kernels.cuh)
//   k_edge_ports_ordered   the port walks of k_edge_ports, one record per start node at its rank among the start nodes
//   k_edge_emit            which ports emit their edge; weights for the three prefix sums (ids, sequence offsets, edge index)
//   k_edge_stamp           per emitted port: walk again, stamp the interior nodes' path words, write the edge's bases and its
//                          record: [0] length | bal_edge << 32, [1] cvg, [2] id, [3] offset of its bases, then the oriented
//                          first and last k-mer (NW words each)
// ===============================================================================================================
constexpr uint64_t GP_SKIP = 1, GP_LINEAR = 2;       // = PATH_SKIP / PATH_LINEAR of sdt_map_kernels.cuh

template <int NW>
__global__ __launch_bounds__(TPB) void k_edge_starts(Table<NW> tbl, const uint64_t *__restrict__ slot_of, uint64_t n, uint32_t *__restrict__ flag,
                                                     uint64_t *__restrict__ pw)
{
	for (uint64_t v = blockIdx.x * (uint64_t)TPB + threadIdx.x; v < n; v += (uint64_t)gridDim.x * TPB) {
		const uint32_t a = tbl.aux[slot_of[v]];
		const bool lin = a & AUX_LINEAR, del = a & AUX_DELETED;
		flag[v] = !lin && !del;
		pw[v] = del ? GP_SKIP : (lin ? (GP_SKIP | GP_LINEAR) : 0ULL);      // skip = deleted || (linear && !inEdge), prlRead2path.c:650
	}
}

struct PortRec { uint64_t far, meta; };              // far node index (~0: no link), length | far_port << 32 | bal_edge << 40

__global__ __launch_bounds__(TPB) void k_edge_start_nodes(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ srank, uint64_t n,
                                                           uint32_t *__restrict__ start_node)
{
	for (uint64_t v = blockIdx.x * (uint64_t)TPB + threadIdx.x; v < n; v += (uint64_t)gridDim.x * TPB)
		if (flag[v]) start_node[srank[v]] = (uint32_t)v;
}

// one lane per PORT of a start node (rank r of the node among the start nodes, port p: record r * 8 + p): every lane of a wave walks
template <int NW>
__global__ __launch_bounds__(TPB) void k_edge_ports_ordered(Table<NW> tbl, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ slot_of,
                                                            const uint32_t *__restrict__ start_node, uint64_t nports, int K,
                                                            uint64_t max_steps, PortRec *__restrict__ ports, Stats *stats)
{
	const Key<NW> mask = key_mask_of<NW>(K);
	const int tb = 2 * (K - 1);
	uint32_t missing = 0;
	for (uint64_t g = blockIdx.x * (uint64_t)TPB + threadIdx.x; g < nports; g += (uint64_t)gridDim.x * TPB) {
		const int p = (int)(g & 7u);
		const Entry<NW> e = tbl.ent[slot_of[start_node[g >> 3]]];
		uint64_t far = ~0ULL, meta = 0;
		const bool live = p < 4 ? ((e.val >> (24 + 6 * p)) & 63u) != 0 : ((e.val >> (6 * (p - 4))) & 63u) != 0;
		if (live) {
			Key<NW> k0;
#pragma unroll
			for (int w = 0; w < NW; w++) k0.w[w] = e.key[w];
			if (p >= 4) k0 = key_revcomp<NW>(k0, K);
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
		ports[g].far = far;
		ports[g].meta = meta;
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}

// one thread per port: does it emit its edge?  (the first of the two ends in visiting order; node2edge.c's zeroing of the far
// link is what keeps the other end quiet.)  asym: a far port whose own walk does not come back here -- the host then builds the
// edges the reference's sequential way.
__global__ __launch_bounds__(TPB) void k_edge_emit(const PortRec *__restrict__ ports, const uint32_t *__restrict__ start_node, const uint32_t *__restrict__ flag,
                                                   const uint32_t *__restrict__ srank, uint64_t nports, uint32_t *__restrict__ w_edge,
                                                   uint32_t *__restrict__ w_id, uint64_t *__restrict__ w_len, unsigned int *asym)
{
	for (uint64_t rp = blockIdx.x * (uint64_t)TPB + threadIdx.x; rp < nports; rp += (uint64_t)gridDim.x * TPB) {
		const PortRec P = ports[rp];
		uint32_t emit = 0;
		if (P.far != ~0ULL) {
			const uint64_t v = start_node[rp >> 3];
			const uint32_t p = (uint32_t)(rp & 7), fp = (uint32_t)(P.meta >> 32) & 0xFFu;
			emit = 1;
			if (flag[P.far]) {
				const PortRec Q = ports[(uint64_t)srank[P.far] * 8 + fp];
				if (Q.far != ~0ULL) {
					if (!(Q.far == v && ((uint32_t)(Q.meta >> 32) & 0xFFu) == p)) atomicOr(asym, 1u);
					emit = (v < P.far || (v == P.far && p <= fp)) ? 1u : 0u;
				}
			}
		}
		w_edge[rp] = emit;
		w_id[rp] = emit ? 1u + ((uint32_t)(P.meta >> 40) & 1u) : 0u;
		w_len[rp] = emit ? (P.meta & 0xFFFFFFFFULL) : 0ULL;
	}
}

template <int NW>
__global__ __launch_bounds__(TPB) void k_edge_stamp(Table<NW> tbl, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ slot_of, int K,
                                                    const PortRec *__restrict__ ports, const uint32_t *__restrict__ start_node, uint64_t nports,
                                                    const uint32_t *__restrict__ w_edge, const uint32_t *__restrict__ e_scan, const uint32_t *__restrict__ id_scan,
                                                    const uint64_t *__restrict__ len_scan, uint64_t *__restrict__ pw, unsigned char *__restrict__ seq,
                                                    uint64_t *__restrict__ erec, Stats *stats)
{
	Key<NW> mask;
#pragma unroll
	for (int i = 0; i < NW; i++) {
		const int bits = 2 * K - 64 * (NW - 1 - i);
		mask.w[i] = bits <= 0 ? 0ULL : (bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL));
	}
	constexpr int RW = 4 + 2 * NW;
	uint32_t missing = 0;
	for (uint64_t rp = blockIdx.x * (uint64_t)TPB + threadIdx.x; rp < nports; rp += (uint64_t)gridDim.x * TPB) {
		if (!w_edge[rp]) continue;
		const PortRec P = ports[rp];
		const uint64_t length = P.meta & 0xFFFFFFFFULL, cnt = length + 1;
		const uint32_t bal = (uint32_t)(P.meta >> 40) & 1u, id = 1u + id_scan[rp];
		const uint64_t e = e_scan[rp], off = len_scan[rp];
		const uint64_t v = start_node[rp >> 3];
		const int p = (int)(rp & 7);
		const Entry<NW> first = tbl.ent[slot_of[v]];
		Key<NW> me;
#pragma unroll
		for (int w = 0; w < NW; w++) me.w[w] = first.key[w];
		const Key<NW> k0 = p < 4 ? me : key_revcomp<NW>(me, K);
		uint32_t b = p < 4 ? (uint32_t)p : ((uint32_t)(p - 4) ^ 2u);
		Key<NW> word = key_next_masked<NW>(k0, b, mask);
		// coverage (merge_linearV2, node2edge.c:474-521): length 1 -- the first node's count; else the four LEFT link counters of
		// every interior node, visited last to first: in a chain that is its own reverse complement a node occurs twice, and its
		// second visit (the position in the lower half) reads the id it was just stamped with -- upstream behaviour, kept
		long long symbol = 0;
		if (length == 1) symbol = (long long)(((uint64_t)(tbl.aux[slot_of[v]] & 0xFFFFu) << 16) | (first.val >> 48));
		const uint32_t idsum = (id & 63u) + ((id >> 6) & 63u) + ((id >> 12) & 63u) + ((id >> 18) & 63u);
		bool ok = true;
		for (uint64_t i = 1; i < cnt; i++) {
			const Key<NW> balk = key_revcomp<NW>(word, K);
			const bool sm = !key_less<NW>(balk, word);
			uint64_t os;
			if (!find_slot<NW>(tbl, sm ? word : balk, os)) { missing++; ok = false; break; }
			seq[off + i - 1] = "ACTG"[word.w[NW - 1] & 3u];
			if (i + 1 == cnt) break;                                     // the last node: not an interior node
			const uint64_t ov = tbl.ent[os].val;
			if (!bal && i < cnt / 2) symbol += idsum;
			else symbol += (long long)((ov & 63u) + ((ov >> 6) & 63u) + ((ov >> 12) & 63u) + ((ov >> 18) & 63u));
			pw[idx[os]] = GP_LINEAR | ((uint64_t)(sm ? bal + 1u : 1u - bal) << 2) | ((uint64_t)(sm ? id : id + bal) << 32);
			b = sm ? first_link((ov >> 24) & 0xFFFFFFu) : (first_link(ov & 0xFFFFFFu) ^ 2u);
			word = key_next_masked<NW>(word, b, mask);
		}
		if (!ok) continue;
		long long cvg = length > 1 ? symbol / (long long)(length - 1) * 10 : symbol / (long long)length * 10;
		if (cvg > 16000) cvg = 16000;                                    // MaxEdgeCov, inc/def.h:37
		uint64_t *R = erec + e * RW;
		R[0] = length | ((uint64_t)bal << 32);
		R[1] = (uint64_t)cvg;
		R[2] = id;
		R[3] = off;
#pragma unroll
		for (int w = 0; w < NW; w++) { R[4 + w] = k0.w[w]; R[4 + NW + w] = word.w[w]; }
	}
	if (missing) atomicAdd(&stats->probe_fail, (unsigned long long)missing);
}


// ===============================================================================================================
// The layout replay ON THE DEVICE (put_kmerset / encap_kmerset, newhash.c:293-462; host twin: graph.c graph_replay_order).
// Per set the reference inserts its distinct keys in first-occurrence order into a table that grows in place.
//   * Between two growths an insertion takes the first free slot from the key's home: first come first served, i.e. the
//     unique layout of priority = insertion rank -- PRIORITY INSERTION (a key that meets a later one takes its slot and
//     carries it on) builds it in any order, all keys of all sets at once (k_rp_put).
//   * A growth re-inserts the old entries in an order that depends on where earlier ones land (an entry that gives way is
//     carried on at once, :359-406).  As a fixed point: an old entry at slot q is inserted at time (q, 0) unless slot q is
//     taken earlier, at time t -- then at t + 1.  From (q, 0) for everybody: lay out by time (priority insertion again,
//     k_rp_insert_timed), read the evictions off the layout (k_rp_times; all times of a round from the layout of the round
//     before), repeat until nothing changes: times only fall, never below the true ones, the only fixed point is the
//     sequential run (tools/replay_fixed_point.c checks exactly this against the sequential emulation).  10-20 rounds.
// Table word: 0 = empty; puts: id + 1; during a rehash: time << qbits | q + 1 with q = the entry's old slot, time = q' << 6 | depth.
// ===============================================================================================================
struct RpSet {
	unsigned long long key0;     // rank of the set's first key in the sorted key array (= its first visiting position)
	unsigned long long tab0;     // first slot of the set's region in the table buffers
	unsigned int m, size, old_size, lo, hi, pad;     // keys; slots now / before the growth; puts insert ids [lo, hi)
};
constexpr int RP_DEPTH_BITS = 6;

__device__ inline int rp_find_set(const unsigned long long *__restrict__ pre, int p, unsigned long long g)
{
	int lo = 0, hi = p;                                // pre[lo] <= g < pre[hi]
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (pre[mid] <= g) lo = mid; else hi = mid;
	}
	return lo;
}

// home slot of a key in a table of `size` slots (newhash.c:423-428 and :43-55; size < 2^32)
template <int NW> __device__ inline uint32_t rp_home(const uint64_t *k, uint32_t size)
{
	if (NW == 1) return (uint32_t)(k[0] % size);
	if (NW == 2) {
		const uint64_t two64 = ((~0ULL) % size + 1ULL) % size;          // 2^64 mod size
		return (uint32_t)(((k[0] % size) * two64 + k[1] % size) % size);
	}
	uint64_t t = k[0] % size;                                        // 32 bits at a time, as modular() does
#pragma unroll
	for (int i = 1; i < NW; i++) {
		t = ((t << 32) | (k[i] >> 32)) % size;
		t = ((t << 32) | (k[i] & 0xFFFFFFFFULL)) % size;
	}
	return (uint32_t)t;
}

// first come first served by priority word (smaller = earlier), any order of execution (Shun & Blelloch's deterministic linear probing)
__device__ inline bool rp_insert(unsigned long long *tab, uint32_t size, uint32_t home, unsigned long long w)
{
	uint32_t i = home;
	for (uint64_t steps = 0; steps < 2ULL * size + 64; steps++) {
		unsigned long long c = __hip_atomic_load(tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (c == 0) {
			c = atomicCAS(tab + i, 0ULL, w);
			if (c == 0) return true;
		}
		if (c > w) {
			const unsigned long long old = atomicCAS(tab + i, c, w);
			if (old != c) continue;                        // somebody else changed the slot: look at it again
			w = c;                                         // the later entry gives way and is carried on
		}
		i = i + 1 == size ? 0u : i + 1;
	}
	return false;
}

// homes of ids [a, b) of every set for a table of size `mod` (pre = exclusive prefix of b - a over the sets)
template <int NW>
__global__ __launch_bounds__(TPB) void k_rp_home(const uint64_t *__restrict__ keys, const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre,
                                                 int p, int rehash, uint32_t *__restrict__ home)
{
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		const unsigned long long id = (rehash ? 0ULL : (unsigned long long)S.lo) + (g - pre[s]);
		home[S.key0 + id] = rp_home<NW>(keys + (S.key0 + id) * NW, S.size);
	}
}

__global__ __launch_bounds__(TPB) void k_rp_put(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int p,
                                                const uint32_t *__restrict__ home, unsigned long long *__restrict__ tab, unsigned int *fail)
{
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		const unsigned long long id = (unsigned long long)S.lo + (g - pre[s]);
		if (!rp_insert(tab + S.tab0, S.size, home[S.key0 + id], id + 1ULL)) atomicOr(fail, 1u);
	}
}

// over the OLD slots of every set (pre = exclusive prefix of old_size; g = pre[s] + q also indexes the per-slot arrays, so a round
// reads times and homes in slot order -- the insertion itself is the only random access of a round)
//   k_rp_rehash_init: time (q, 0) and the home in the NEW geometry of the entry at every old slot
template <int NW>
__global__ __launch_bounds__(TPB) void k_rp_rehash_init(const uint64_t *__restrict__ keys, const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre,
                                                        int p, const unsigned long long *__restrict__ told, unsigned long long *__restrict__ t_slot,
                                                        uint32_t *__restrict__ home_slot)
{
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		const unsigned long long q = g - pre[s];
		const unsigned long long y = told[S.tab0 + q];               // id + 1
		if (!y) continue;
		t_slot[g] = q << RP_DEPTH_BITS;
		home_slot[g] = rp_home<NW>(keys + (S.key0 + y - 1ULL) * NW, S.size);
	}
}

//   The rounds of a growth (table word here: time << qbits | q + 1, q = the entry's OLD slot; home and time are per old slot).
//   The set of occupied slots of a linear-probing table does not depend on the order of insertion, and the word at slot i depends on
//   the entries with a home at or before i only (probing goes forward): an entry of home h whose time changed can re-arrange the slots
//   from h to the end of its cluster (the next empty slot) and nothing else.  Round 0 is full (k_rp_ins_all, k_rp_eval_all); every later
//   round takes those stretches out of the table (k_rp_collect: walkers that meet share the work -- whoever empties a slot goes on to
//   the next one, so every slot up to the end of the cluster is taken exactly once), re-inserts their entries with the new times
//   (k_rp_ins_list: an entry of an earlier home passes over the untouched slots before h, all of them hold earlier entries) and
//   re-evaluates the old slots that lie inside them (k_rp_eval_list: an evaluation depends on the word at the entry's old slot and on
//   its own time only, and is idempotent) -- about half of one full round in all instead of 15-20 full rounds (tools/replay_fixed_point.c,
//   table C).  List entry: set << 54 | length taken << 32 | first slot.
constexpr int RP_LEN_BITS = 22;
struct RpRound {
	unsigned long long n_next;       // CHUNKS of the list this round writes (sdt_append.cuh: unused slots are marked AP_NONE)
	unsigned int flags;              // 2 = an insertion found no slot, 4 = chain or stretch past its field, 8 = the list is full
	unsigned int pad;
};

__global__ __launch_bounds__(TPB) void k_rp_ins_all(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int p, int qbits,
                                                    const unsigned long long *__restrict__ told, unsigned long long *tnew,
                                                    const uint32_t *__restrict__ home_slot, const unsigned long long *__restrict__ t, RpRound *st)
{
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		const unsigned long long q = g - pre[s];
		if (!told[S.tab0 + q]) continue;
		if (!rp_insert(tnew + S.tab0, S.size, home_slot[g], (t[g] << qbits) | (q + 1ULL))) atomicOr(&st->flags, 2u);
	}
}

// next time of the entry of old slot q (g = its index over all sets) from the layout; a change lists the entry's home
__device__ inline void rp_eval(const RpSet &S, int s, unsigned long long g, unsigned long long q, int qbits, const unsigned long long *tnew,
                               const uint32_t *__restrict__ home_slot, unsigned long long *t, unsigned long long *next_list, WaveApp *app,
                               const ApOut &out, RpRound *st)
{
	const unsigned long long w = tnew[S.tab0 + q], x = w & ((1ULL << qbits) - 1ULL), tx = w >> qbits, mine = t[g], scan = q << RP_DEPTH_BITS;
	unsigned long long nt = scan;
	if (x == q + 1ULL) nt = mine;                                  // it sits on its own old slot: nobody took it (leave the time alone)
	else if (w && tx < scan) nt = tx + 1ULL;                       // the slot was taken before the scan reached it: carried on at once
	if (nt == mine) return;
	if ((nt & ((1ULL << RP_DEPTH_BITS) - 1ULL)) == (1ULL << RP_DEPTH_BITS) - 1ULL) atomicOr(&st->flags, 4u);      // chain too deep for the field
	t[g] = nt;
	const unsigned long long k = ap_append(app, out);
	if (k != AP_NONE) next_list[k] = ((unsigned long long)s << (32 + RP_LEN_BITS)) | home_slot[g];
	else atomicOr(&st->flags, 8u);
}

__global__ __launch_bounds__(TPB) void k_rp_eval_all(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int p, int qbits,
                                                     const unsigned long long *__restrict__ told, const unsigned long long *tnew,
                                                     const uint32_t *__restrict__ home_slot, unsigned long long *t,
                                                     unsigned long long *next_list, unsigned long long cap_chunks, RpRound *st)
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const ApOut out = {&st->n_next, cap_chunks, nullptr, next_list};
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		const unsigned long long q = g - pre[s];
		if (!told[S.tab0 + q]) continue;
		rp_eval(S, s, g, q, qbits, tnew, home_slot, t, next_list, s_app, out, st);
	}
	ap_finish_mark(s_app, out);
}

// take the listed stretches out of the table: the words go to saved[] (same index as the slot), the slots are emptied
__global__ __launch_bounds__(TPB) void k_rp_collect(const RpSet *__restrict__ sets, unsigned long long *tnew, unsigned long long *__restrict__ saved,
                                                    unsigned long long *list, unsigned long long n_list, RpRound *st)
{
	for (unsigned long long k = blockIdx.x * (unsigned long long)TPB + threadIdx.x; k < n_list; k += (unsigned long long)gridDim.x * TPB) {
		const unsigned long long e = list[k];
		if (e == AP_NONE) continue;
		const RpSet S = sets[(int)(e >> (32 + RP_LEN_BITS))];
		unsigned long long *T = tnew + S.tab0;
		uint32_t len = 0, i = (uint32_t)e;
		while (len < S.size) {
			if (__hip_atomic_load(T + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;      // (the end, or a walker ahead of this one)
			const unsigned long long w = atomicExch(T + i, 0ULL);
			if (w == 0) break;
			saved[S.tab0 + i] = w;
			len++;
			i = i + 1u == S.size ? 0u : i + 1u;
		}
		if (len >= (1u << RP_LEN_BITS)) { atomicOr(&st->flags, 4u); len = 0; }
		list[k] = e | ((unsigned long long)len << 32);
	}
}

__global__ __launch_bounds__(TPB) void k_rp_ins_list(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int qbits,
                                                     unsigned long long *tnew, const unsigned long long *__restrict__ saved,
                                                     const uint32_t *__restrict__ home_slot, const unsigned long long *__restrict__ t,
                                                     const unsigned long long *__restrict__ list, unsigned long long n_list, RpRound *st)
{
	const unsigned long long qmask = (1ULL << qbits) - 1ULL;
	for (unsigned long long k = blockIdx.x * (unsigned long long)TPB + threadIdx.x; k < n_list; k += (unsigned long long)gridDim.x * TPB) {
		const unsigned long long e = list[k];
		if (e == AP_NONE) continue;
		const int s = (int)(e >> (32 + RP_LEN_BITS));
		const uint32_t len = (uint32_t)(e >> 32) & ((1u << RP_LEN_BITS) - 1u);
		const RpSet S = sets[s];
		uint32_t i = (uint32_t)e;
		for (uint32_t j = 0; j < len; j++) {
			const unsigned long long q1 = saved[S.tab0 + i] & qmask, g = pre[s] + q1 - 1ULL;
			if (!rp_insert(tnew + S.tab0, S.size, home_slot[g], (t[g] << qbits) | q1)) atomicOr(&st->flags, 2u);
			i = i + 1u == S.size ? 0u : i + 1u;
		}
	}
}

// the old slots inside the re-arranged stretches are the only ones whose evaluation can have changed
__global__ __launch_bounds__(TPB) void k_rp_eval_list(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int qbits,
                                                      const unsigned long long *__restrict__ told, const unsigned long long *tnew,
                                                      const uint32_t *__restrict__ home_slot, unsigned long long *t,
                                                      const unsigned long long *__restrict__ list, unsigned long long n_list,
                                                      unsigned long long *next_list, unsigned long long cap_chunks, RpRound *st)
{
	__shared__ WaveApp s_app[TPB / 64];
	ap_init(s_app);
	const ApOut out = {&st->n_next, cap_chunks, nullptr, next_list};
	for (unsigned long long k = blockIdx.x * (unsigned long long)TPB + threadIdx.x; k < n_list; k += (unsigned long long)gridDim.x * TPB) {
		const unsigned long long e = list[k];
		if (e == AP_NONE) continue;
		const int s = (int)(e >> (32 + RP_LEN_BITS));
		const uint32_t len = (uint32_t)(e >> 32) & ((1u << RP_LEN_BITS) - 1u);
		const RpSet S = sets[s];
		uint32_t i = (uint32_t)e;
		for (uint32_t j = 0; j < len; j++) {
			if (i < S.old_size && told[S.tab0 + i]) rp_eval(S, s, pre[s] + i, i, qbits, tnew, home_slot, t, next_list, s_app, out, st);
			i = i + 1u == S.size ? 0u : i + 1u;
		}
	}
	ap_finish_mark(s_app, out);
}

// over the NEW slots (pre = exclusive prefix of size): mode 0 = after a growth: the old slot in the word -> the entry (id + 1, from the
// old table), 1 = flag occupied slots for the final order
__global__ __launch_bounds__(TPB) void k_rp_slots(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int p, int mode, int qbits,
                                                  const unsigned long long *__restrict__ told, unsigned long long *__restrict__ tab, uint32_t *__restrict__ occ)
{
	const unsigned long long total = pre[p], qmask = (1ULL << qbits) - 1ULL;
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		const int s = rp_find_set(pre, p, g);
		const unsigned long long at = sets[s].tab0 + (g - pre[s]);
		if (mode == 0) {
			const unsigned long long w = tab[at];
			if (w) tab[at] = told[sets[s].tab0 + (w & qmask) - 1ULL];
		} else occ[g] = tab[at] != 0;
	}
}

// order[v] = rank of the node at visiting position v: the sets one after the other, inside a set the table's slots in order
__global__ __launch_bounds__(TPB) void k_rp_order(const RpSet *__restrict__ sets, const unsigned long long *__restrict__ pre, int p,
                                                  const unsigned long long *__restrict__ tab, const uint32_t *__restrict__ occ,
                                                  const uint32_t *__restrict__ rank, uint64_t *__restrict__ order)
{
	const unsigned long long total = pre[p];
	for (unsigned long long g = blockIdx.x * (unsigned long long)TPB + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * TPB) {
		if (!occ[g]) continue;
		const int s = rp_find_set(pre, p, g);
		const RpSet S = sets[s];
		// rank[] is the exclusive scan of occ over all sets' slots: the entries before this set's region are exactly its key0
		order[rank[g]] = S.key0 + (tab[S.tab0 + (g - pre[s])] - 1ULL);
	}
}


// ---- the arcs of the second read pass in the order of *.preArc (output_arcs, prlRead2path.c:454-505) --------------------------
__global__ __launch_bounds__(TPB) void k_arc_keys(const uint64_t *__restrict__ first, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ perm)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) { key[i] = ~first[i]; perm[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(TPB) void k_arc_from_of(const uint32_t *__restrict__ from, const uint32_t *__restrict__ perm, uint64_t n, uint32_t *__restrict__ key)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) key[i] = from[perm[i]];
}
__global__ __launch_bounds__(TPB) void k_arc_gather(const uint32_t *__restrict__ perm, uint64_t n, const uint32_t *__restrict__ f, const uint32_t *__restrict__ t,
                                                    const uint32_t *__restrict__ m, const uint64_t *__restrict__ o, uint32_t *__restrict__ f2,
                                                    uint32_t *__restrict__ t2, uint32_t *__restrict__ m2, uint64_t *__restrict__ o2)
{
	for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
		const uint32_t j = perm[i];
		f2[i] = f[j]; t2[i] = t[j]; m2[i] = m[j]; o2[i] = o[j];
	}
}
