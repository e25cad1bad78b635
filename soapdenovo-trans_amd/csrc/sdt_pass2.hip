// sdt_pass2.hip -- the second pass over the reads (prlRead2edge, prlRead2path.c:817-1335) on the device: the cleaned graph comes back as
// one path word per node + a patch table of the length-1 edges, k_map_reads threads every kept read through it, the arcs go out in
// *.preArc's order.
#include "sdt_ctx.hpp"
#include "sdt_append.cuh"
#include "sdt_map_kernels.cuh"
#include "sdt_table_kernels.cuh"

template <int NW>
static int build_patch_table(sdt_ctx *c, const uint64_t *pkeys, const uint64_t *pinfo, uint64_t np)
{
	uint64_t slots = 1024;
	while (slots < 2 * np + 2)
		slots <<= 1;
	std::vector<PatchEnt<NW>> tab(slots);
	for (auto &e : tab) {
		for (int w = 0; w < NW; w++) e.key[w] = KEY_EMPTY;
		e.info = 0;
	}
	for (uint64_t i = 0; i < np; i++) {
		Key<NW> k;
		for (int w = 0; w < NW; w++) k.w[w] = pkeys[i * NW + w];
		uint64_t s = key_hash<NW>(k) & (slots - 1);
		while (tab[s].key[0] != KEY_EMPTY) s = (s + 1) & (slots - 1);
		for (int w = 0; w < NW; w++) tab[s].key[w] = k.w[w];
		tab[s].info = pinfo[i];
	}
	if (c->d_patch) HIPCHK(hipFree(c->d_patch));
	c->d_patch = nullptr;
	HIPCHK(hipMalloc(&c->d_patch, slots * sizeof(PatchEnt<NW>)));
	HIPCHK(hipMemcpy(c->d_patch, tab.data(), slots * sizeof(PatchEnt<NW>), hipMemcpyHostToDevice));
	c->patch_slots = slots;
	return SDT_OK;
}

extern "C" {
// ---- second pass: prlRead2edge on the device ---------------------------------------------------------------
int sdt_gpu_load_paths(sdt_ctx *c, const uint64_t *keys, const uint64_t *path_words, uint64_t n, const uint64_t *patch_keys,
                       const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed)
{
	if (!c || (npatch && (!patch_keys || !patch_info)))
		return fail(SDT_EINVAL, "NULL argument");
	const bool by_index = keys == nullptr;
	// keys == NULL and path_words == NULL: the path words sdt_gpu_build_edges left on the device
	uint64_t *d_own = (!keys && !path_words && n) ? sdti::graph_take_path_words(c->gx, n) : nullptr;
	if (n && !path_words && !d_own)
		return fail(keys ? SDT_EINVAL : SDT_ESTATE, "no path words: pass them, or build the edges with sdt_gpu_build_edges first");
	if (by_index && n && (!c->d_idx || c->idx_slots != view_slots(c) || c->idx_n != n)) {
		if (d_own) (void)hipFree(d_own);
		return fail(SDT_ESTATE, "keys == NULL needs the node index of sdt_gpu_set_node_index for the same %llu nodes", (unsigned long long)n);
	}
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	uint64_t *d_k = nullptr, *d_i = nullptr;
	if (n) {
		if (!by_index) HIPCHK(hipMalloc((void **)&d_k, n * c->nw * sizeof(uint64_t)));
		hipError_t e = d_own ? hipSuccess : hipMalloc((void **)&d_i, n * sizeof(uint64_t));
		if (e != hipSuccess) { if (d_k) (void)hipFree(d_k); return fail(SDT_ENOMEM, "path words: %s", hipGetErrorString(e)); }
		if (d_own) d_i = d_own;
		int rcu = by_index ? SDT_OK : sdti::h2d_big(c->copy_stream, d_k, keys, n * c->nw * sizeof(uint64_t));
		if (rcu == SDT_OK && !d_own) rcu = sdti::h2d_big(c->copy_stream, d_i, path_words, n * sizeof(uint64_t));
		if (rcu != SDT_OK) { if (d_k) (void)hipFree(d_k); (void)hipFree(d_i); return rcu; }
		if (by_index) {
			const int g = scan_grid(c, view_slots(c));
			if (c->nw == 1) hipLaunchKernelGGL(k_set_paths_by_index<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), c->d_idx, d_i, n, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_set_paths_by_index<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), c->d_idx, d_i, n, c->d_stats);
			else hipLaunchKernelGGL(k_set_paths_by_index<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), c->d_idx, d_i, n, c->d_stats);
		} else {
			const int g = scan_grid(c, n);
			if (c->nw == 1) hipLaunchKernelGGL(k_set_paths<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_k, d_i, n, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_set_paths<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_k, d_i, n, c->d_stats);
			else hipLaunchKernelGGL(k_set_paths<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_k, d_i, n, c->d_stats);
		}
		hipError_t le = hipGetLastError();
		hipError_t se = hipStreamSynchronize(c->stream);
		if (d_k) (void)hipFree(d_k);
		(void)hipFree(d_i);
		if (le != hipSuccess || se != hipSuccess)
			return fail(SDT_EHIP, "k_set_paths: %s", hipGetErrorString(le != hipSuccess ? le : se));
	}
	int rc = c->nw == 1 ? build_patch_table<1>(c, patch_keys, patch_info, npatch)
	       : c->nw == 2 ? build_patch_table<2>(c, patch_keys, patch_info, npatch)
	                    : build_patch_table<4>(c, patch_keys, patch_info, npatch);
	if (rc != SDT_OK)
		return rc;
	// arcs: a few per edge in practice; the map doubles (and the pass is redone) if it ever fills up
	uint64_t slots = 1 << 16;
	while (slots < 8 * (num_ed + 1))
		slots <<= 1;
	if (c->d_arcs) HIPCHK(hipFree(c->d_arcs));
	c->d_arcs = nullptr;
	HIPCHK(hipMalloc((void **)&c->d_arcs, slots * sizeof(ArcEnt)));
	c->arc_slots = slots;
	rc = sync_stats(c);
	if (rc != SDT_OK)
		return fail(SDT_ESTATE, "sdt_gpu_load_paths: %llu nodes are not in the table", (unsigned long long)c->h_stats->probe_fail);
	c->paths_loaded = true;
	return SDT_OK;
}

int sdt_gpu_export_paths(sdt_ctx *c, uint64_t *keys, uint64_t *path_words, uint64_t max_nodes, uint64_t *n)
{
	if (!c || !n)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->paths_loaded)
		return fail(SDT_ESTATE, "call sdt_gpu_load_paths first");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	const uint64_t nodes = c->h_stats->distinct;
	*n = nodes;
	if (!keys && !path_words)
		return SDT_OK;
	if (!keys || !path_words || max_nodes < nodes)
		return fail(SDT_EINVAL, "export arrays hold %llu nodes, the table has %llu", (unsigned long long)max_nodes, (unsigned long long)nodes);
	uint64_t *d_k = nullptr, *d_p = nullptr;
	const uint64_t m = nodes ? nodes : 1;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	hipError_t e = hipMalloc((void **)&d_p, m * 8);
	if (e != hipSuccess) { (void)hipFree(d_k); return fail(SDT_ENOMEM, "path export: %s", hipGetErrorString(e)); }
	int ret = SDT_OK;
	e = hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream);
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_export_paths<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_export_paths<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	else hipLaunchKernelGGL(k_export_paths<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_k, d_p, (unsigned long long)nodes, c->d_stats);
	if (e == hipSuccess) e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) ret = fail(SDT_EHIP, "k_export_paths: %s", hipGetErrorString(e));
	if (ret == SDT_OK) {
		// every non-empty slot took a place: a table whose counters disagree with its slots would publish an incomplete graph
		unsigned long long placed = 0;
		e = hipMemcpy(&placed, &c->d_stats->scratch, sizeof placed, hipMemcpyDeviceToHost);
		if (e != hipSuccess) ret = fail(SDT_EHIP, "k_export_paths: %s", hipGetErrorString(e));
		else if (placed != nodes)
			ret = fail(SDT_ESTATE, "path export: the table holds %llu nodes, its counters say %llu", placed, (unsigned long long)nodes);
	}
	if (ret == SDT_OK) ret = sdti::d2h_big(c->copy_stream, keys, d_k, nodes * c->nw * 8);
	if (ret == SDT_OK) ret = sdti::d2h_big(c->copy_stream, path_words, d_p, nodes * 8);
	(void)hipFree(d_k);
	(void)hipFree(d_p);
	return ret;
}

int sdt_gpu_import_paths(sdt_ctx *c, const uint64_t *keys, const uint64_t *path_words, uint64_t n, const uint64_t *patch_keys,
                         const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed)
{
	if (!c || (n && (!keys || !path_words)) || (npatch && (!patch_keys || !patch_info)))
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	// the table of this rank's shard makes way (the reads kept for the second pass stay): an empty flat table with room for the graph
	if (c->d_idx) { (void)hipFree(c->d_idx); c->d_idx = nullptr; c->idx_slots = c->idx_n = 0; }
	const uint64_t want = flat_slots_for(n);
	if (want > c->slots) {
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->d_ent) (void)hipFree(c->d_ent);
		if (c->d_aux) (void)hipFree(c->d_aux);
		if (c->d_first) (void)hipFree(c->d_first);
		c->d_ent = nullptr; c->d_aux = nullptr; c->d_first = nullptr;
		rc = alloc_table(c, want, &c->d_ent, &c->d_aux, &c->d_first);
		if (rc != SDT_OK) { c->slots = 0; return rc; }
		c->slots = want;
	}
	rc = launch_clear(c, c->d_ent, c->d_aux, c->d_first, c->slots);
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemsetAsync(&c->d_stats->distinct, 0, sizeof(unsigned long long), c->stream));
	c->distinct_known = 0;
	c->kmers_since_sync = c->hard_since_sync = 0;
	uint64_t *d_k = nullptr, *d_p = nullptr;
	const uint64_t STEP = 1ULL << 26;                // nodes per upload: bounded staging memory
	const uint64_t m = n < STEP ? (n ? n : 1) : STEP;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	hipError_t e = hipMalloc((void **)&d_p, m * 8);
	if (e != hipSuccess) { (void)hipFree(d_k); return fail(SDT_ENOMEM, "path import: %s", hipGetErrorString(e)); }
	for (uint64_t i0 = 0; i0 < n && rc == SDT_OK; i0 += STEP) {
		const uint64_t k = n - i0 < STEP ? n - i0 : STEP;
		rc = sdti::h2d_big(c->copy_stream, d_k, keys + i0 * c->nw, k * c->nw * 8);
		if (rc == SDT_OK) rc = sdti::h2d_big(c->copy_stream, d_p, path_words + i0, k * 8);
		if (rc != SDT_OK) break;
		const int g = scan_grid(c, k);
		if (c->nw == 1) hipLaunchKernelGGL(k_import_paths<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d_k, d_p, k, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_import_paths<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d_k, d_p, k, c->d_stats);
		else hipLaunchKernelGGL(k_import_paths<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d_k, d_p, k, c->d_stats);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e != hipSuccess) rc = fail(SDT_EHIP, "k_import_paths: %s", hipGetErrorString(e));
	}
	(void)hipFree(d_k);
	(void)hipFree(d_p);
	if (rc != SDT_OK) return rc;
	// patch table, arc map, and the check that every key went in once (sdt_gpu_load_paths with no node of its own to set)
	rc = sdt_gpu_load_paths(c, nullptr, nullptr, 0, patch_keys, patch_info, npatch, num_ed);
	if (rc == SDT_OK && c->h_stats->distinct != n)
		return fail(SDT_ESTATE, "path import: %llu keys came, the table holds %llu nodes (a key twice?)", (unsigned long long)n,
		            (unsigned long long)c->h_stats->distinct);
	return rc;
}
int sdt_gpu_map_reads(sdt_ctx *c, uint64_t *reads_processed, uint64_t *arcs)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (!c->paths_loaded)
		return fail(SDT_ESTATE, "call sdt_gpu_load_paths first");
	if (!(c->flags & SDT_FLAG_KEEP_READS) && c->kept.empty())
		return fail(SDT_ESTATE, "the reads were not kept: init with SDT_FLAG_KEEP_READS (or hand them over with sdt_gpu_keep_reads)");
	HIPCHK(hipSetDevice(c->device));
	for (int attempt = 0; attempt < 8; attempt++) {
		// ArcEnt.first starts at ~0 (atomicMin), key/mult at 0
		HIPCHK(hipMemsetAsync(c->d_arcs, 0, c->arc_slots * sizeof(ArcEnt), c->stream));
		HIPCHK(hipMemset2DAsync(&c->d_arcs[0].first, sizeof(ArcEnt), 0xFF, sizeof(unsigned long long), c->arc_slots, c->stream));
		HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
		uint64_t reads = 0;
		// One lane per read and ~120 dependent look-ups per lane: a kept batch of the CLI (10^5 reads) is 1 600 waves, six per CU, and
		// its launch lasts as long as the longest chain (0.4 ms: 1 900 launches one after the other took 770 ms at 200 M reads).  The
		// batches are independent (arcs are atomic adds / mins): several streams keep several launches on the device at a time.
		constexpr int NS = 6;
		hipStream_t ms[NS];
		hipEvent_t ready, done[NS];
		int ns = c->kept.size() > 8 ? NS : 1;
		if (ns > 1) {
			if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) ns = 1;
			for (int i = 0; i < ns && ns > 1; i++)
				if (hipStreamCreateWithFlags(&ms[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
					return fail(SDT_EHIP, "streams for the second read pass");
		}
		if (ns > 1) {
			HIPCHK(hipEventRecord(ready, c->stream));
			for (int i = 0; i < ns; i++) HIPCHK(hipStreamWaitEvent(ms[i], ready, 0));
		}
		size_t bi = 0;
		for (auto &kb : c->kept) {
			const int g = scan_grid(c, kb.nreads);
			const hipStream_t st = ns > 1 ? ms[bi++ % (size_t)ns] : c->stream;
			if (c->nw == 1) hipLaunchKernelGGL(k_map_reads<1>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<1>(c), (const PatchEnt<1> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			else if (c->nw == 2) hipLaunchKernelGGL(k_map_reads<2>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<2>(c), (const PatchEnt<2> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			else hipLaunchKernelGGL(k_map_reads<4>, dim3(g), dim3(TPB), 0, st, kb.d_words, kb.d_offs, kb.nreads, c->K, table_of<4>(c), (const PatchEnt<4> *)c->d_patch, c->patch_slots - 1, c->d_arcs, c->arc_slots - 1, kb.ord_base, kb.ord_stride, c->d_stats);
			HIPCHK(hipGetLastError());
			reads += kb.nreads;
		}
		if (ns > 1) {
			for (int i = 0; i < ns; i++) { HIPCHK(hipEventRecord(done[i], ms[i])); HIPCHK(hipStreamWaitEvent(c->stream, done[i], 0)); }
		}
		HIPCHK(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (ns > 1) {
			for (int i = 0; i < ns; i++) { (void)hipStreamDestroy(ms[i]); (void)hipEventDestroy(done[i]); }
			(void)hipEventDestroy(ready);
		}
		if (c->h_stats->scratch)
			return fail(SDT_ESTATE, "%llu reads hold a k-mer that is not in the node table (different reads than pass 1?)",
			            (unsigned long long)c->h_stats->scratch);
		if (c->h_stats->probe_fail == 0) {
			if (reads_processed) *reads_processed = reads;
			if (arcs) {
				// count occupied slots by exporting nothing but the cursor
				unsigned long long *d_cur = nullptr;
				HIPCHK(hipMalloc((void **)&d_cur, sizeof(unsigned long long)));
				HIPCHK(hipMemsetAsync(d_cur, 0, sizeof(unsigned long long), c->stream));
				hipLaunchKernelGGL(k_export_arcs, dim3(scan_grid(c, c->arc_slots)), dim3(TPB), 0, c->stream, c->d_arcs, c->arc_slots, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, 0ULL, d_cur);
				unsigned long long h = 0;
				HIPCHK(hipMemcpyAsync(&h, d_cur, sizeof h, hipMemcpyDeviceToHost, c->stream));
				HIPCHK(hipStreamSynchronize(c->stream));
				(void)hipFree(d_cur);
				*arcs = h;
			}
			return SDT_OK;
		}
		// arc map too small: double it and redo the pass (arc adds are idempotent only from a clean map)
		HIPCHK(hipMemsetAsync(&c->d_stats->probe_fail, 0, sizeof(unsigned long long), c->stream));
		HIPCHK(hipFree(c->d_arcs));
		c->d_arcs = nullptr;
		c->arc_slots <<= 1;
		HIPCHK(hipMalloc((void **)&c->d_arcs, c->arc_slots * sizeof(ArcEnt)));
	}
	return fail(SDT_EFULL, "arc map keeps overflowing");
}

int sdt_gpu_export_arcs(sdt_ctx *c, uint32_t *from, uint32_t *to, uint32_t *mult, uint64_t *first, uint64_t max_arcs, uint64_t *n)
{
	if (!c || !from || !to || !mult || !first)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->d_arcs)
		return fail(SDT_ESTATE, "no arcs: call sdt_gpu_map_reads first");
	HIPCHK(hipSetDevice(c->device));
	uint32_t *d_f = nullptr, *d_t = nullptr, *d_m = nullptr;
	uint64_t *d_o = nullptr;
	unsigned long long *d_cur = nullptr;
	const uint64_t m = max_arcs ? max_arcs : 1;
	int ret = SDT_OK;
	unsigned long long h = 0;
#define ARC_CHK(expr) do { hipError_t e4_ = (expr); if (e4_ != hipSuccess) { ret = fail(SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e4_)); goto done; } } while (0)
	ARC_CHK(hipMalloc((void **)&d_f, m * 4));
	ARC_CHK(hipMalloc((void **)&d_t, m * 4));
	ARC_CHK(hipMalloc((void **)&d_m, m * 4));
	ARC_CHK(hipMalloc((void **)&d_o, m * 8));
	ARC_CHK(hipMalloc((void **)&d_cur, 8));
	ARC_CHK(hipMemsetAsync(d_cur, 0, 8, c->stream));
	hipLaunchKernelGGL(k_export_arcs, dim3(scan_grid(c, c->arc_slots)), dim3(TPB), 0, c->stream, c->d_arcs, c->arc_slots, d_f, d_t, d_m, d_o, (unsigned long long)max_arcs, d_cur);
	ARC_CHK(hipGetLastError());
	ARC_CHK(hipMemcpyAsync(&h, d_cur, 8, hipMemcpyDeviceToHost, c->stream));
	ARC_CHK(hipStreamSynchronize(c->stream));
	if (h > max_arcs) { ret = fail(SDT_EINVAL, "arc arrays hold %llu, need %llu", (unsigned long long)max_arcs, h); goto done; }
	// (in the order *.preArc lists them: the host's own sort finds nothing left to do)
	ret = sdti::sort_arcs_for_output(c->stream, c->cu_count, d_f, d_t, d_m, d_o, h);
	if (ret != SDT_OK) goto done;
	ARC_CHK(hipMemcpy(from, d_f, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(to, d_t, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(mult, d_m, h * 4, hipMemcpyDeviceToHost));
	ARC_CHK(hipMemcpy(first, d_o, h * 8, hipMemcpyDeviceToHost));
	if (n) *n = h;
done:
	if (d_f) (void)hipFree(d_f);
	if (d_t) (void)hipFree(d_t);
	if (d_m) (void)hipFree(d_m);
	if (d_o) (void)hipFree(d_o);
	if (d_cur) (void)hipFree(d_cur);
	return ret;
}
int sdt_gpu_keep_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads)
{
	if (!c || !packed_words || !offsets)
		return fail(SDT_EINVAL, "NULL argument");
	if (nreads == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++)
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	if (((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short");
	sdt_ctx::KeptBatch kb;
	kb.nwords = nwords; kb.nreads = nreads; kb.ord_base = c->ord_base; kb.ord_stride = c->ord_stride; kb.maxlen = maxlen;
	kb.d_words = (uint32_t *)keep_alloc(c, nwords * sizeof(uint32_t));
	kb.d_offs = (uint64_t *)keep_alloc(c, (nreads + 1) * sizeof(uint64_t));
	if (!kb.d_words || !kb.d_offs)
		return fail(SDT_ENOMEM, "kept reads: no device memory for another batch (%zu slabs held); run with --host-map", c->keep_slabs.size());
	c->kept.push_back(kb);
	HIPCHK(hipMemcpyAsync(kb.d_words, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipMemcpyAsync(kb.d_offs, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipStreamSynchronize(c->copy_stream));
	return SDT_OK;
}
} // extern "C"
