// sdt_mapstage.hip -- the map stage's hashing path: prlContig2nodes (prlHashCtg.c:287-425) = k_index_contigs,
// prlRead2Ctg (prlRead2Ctg.c:656-894) = k_align_reads.
#include "sdt_ctx.hpp"
#include "sdt_ctg_kernels.cuh"

extern "C" {
// ---- map stage: prlContig2nodes / prlRead2Ctg ----------------------------------------------------------------
static int ab_reserve(sdt_ctx *c, int i, size_t bytes)
{
	if (c->ab_cap[i] >= bytes) return SDT_OK;
	if (c->ab[i]) HIPCHK(hipFree(c->ab[i]));
	c->ab[i] = nullptr;
	c->ab_cap[i] = 0;
	const size_t want = bytes + bytes / 4 + 256;
	hipError_t e = hipMalloc(&c->ab[i], want);
	if (e != hipSuccess) return fail(SDT_ENOMEM, "map staging (%zu bytes): %s", want, hipGetErrorString(e));
	c->ab_cap[i] = want;
	return SDT_OK;
}

int sdt_gpu_index_contigs(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, const uint32_t *ids,
                          uint64_t ncontigs)
{
	if (!c || !packed_words || !offsets || !ids)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX to index contigs");
	if (c->index_final)
		return fail(SDT_ESTATE, "the contig index is final once reads have been aligned: sdt_gpu_reset to start over");
	if (ncontigs == 0)
		return SDT_OK;
	uint64_t kmers = 0;
	for (uint64_t i = 0; i < ncontigs; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at contig %llu", (unsigned long long)i);
		const uint64_t len = offsets[i + 1] - offsets[i];
		if (len >= (1ULL << CTG_POS_BITS))
			return fail(SDT_EINVAL, "contig %llu is %llu bases long: positions are 24-bit (kmer_t.r_links)", (unsigned long long)i, (unsigned long long)len);
		if (len >= (uint64_t)c->K) kmers += len - c->K + 1;
	}
	if (((offsets[ncontigs] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words", (unsigned long long)(((offsets[ncontigs] + 15) >> 4) + TAIL_PAD), TAIL_PAD);
	HIPCHK(hipSetDevice(c->device));
	// contig ordinal -> id table grows by this batch
	if (c->ctg_ord + ncontigs > c->ctg_ids_cap) {
		const uint64_t cap = (c->ctg_ord + ncontigs) * 2 + 1024;
		uint32_t *n = nullptr;
		HIPCHK(hipMalloc((void **)&n, cap * sizeof(uint32_t)));
		if (c->ctg_ord) HIPCHK(hipMemcpyAsync(n, c->d_ctg_ids, c->ctg_ord * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->d_ctg_ids) HIPCHK(hipFree(c->d_ctg_ids));
		c->d_ctg_ids = n;
		c->ctg_ids_cap = cap;
	}
	int rc = ab_reserve(c, 0, nwords * sizeof(uint32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 1, (ncontigs + 1) * sizeof(uint64_t));
	if (rc != SDT_OK) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[0], packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[1], offsets, (ncontigs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->d_ctg_ids + c->ctg_ord, ids, ncontigs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	rc = ensure_room(c, kmers);
	if (rc != SDT_OK) return rc;
	const int g = scan_grid(c, offsets[ncontigs] ? offsets[ncontigs] : 1);
	const uint32_t *dw = (const uint32_t *)c->ab[0];
	const uint64_t *dof = (const uint64_t *)c->ab[1];
	if (c->nw == 1) hipLaunchKernelGGL(k_index_contigs<1>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<1>(c), c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_index_contigs<2>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<2>(c), c->d_stats);
	else hipLaunchKernelGGL(k_index_contigs<4>, dim3(g), dim3(TPB), 0, c->stream, dw, dof, ncontigs, c->ctg_ord, c->K, table_of<4>(c), c->d_stats);
	HIPCHK(hipGetLastError());
	c->kmers_since_sync += kmers;
	c->ctg_ord += ncontigs;
	HIPCHK(hipStreamSynchronize(c->stream));     // the caller may reuse its buffers
	return SDT_OK;
}

int sdt_gpu_set_contig_table(sdt_ctx *c, const uint32_t *length, const uint32_t *twin, uint64_t num_ctg)
{
	if (!c || !length || !twin)
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->d_ctg_len) HIPCHK(hipFree(c->d_ctg_len));
	if (c->d_ctg_twin) HIPCHK(hipFree(c->d_ctg_twin));
	c->d_ctg_len = c->d_ctg_twin = nullptr;
	HIPCHK(hipMalloc((void **)&c->d_ctg_len, (num_ctg + 1) * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&c->d_ctg_twin, (num_ctg + 1) * sizeof(uint32_t)));
	HIPCHK(hipMemcpy(c->d_ctg_len, length, (num_ctg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(c->d_ctg_twin, twin, (num_ctg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
	c->num_ctg = num_ctg;
	return SDT_OK;
}

static int launch_align(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads, uint64_t max_read_len,
                        const int32_t *d_align_len, int align_len_all, uint64_t *d_info, Hit *d_hits, uint64_t max_hits)
{
	if (!c->d_ctg_len)
		return fail(SDT_ESTATE, "call sdt_gpu_set_contig_table first");
	if (max_read_len < (uint64_t)c->K + 1) max_read_len = (uint64_t)c->K + 1;
	const int max_kmers = (int)(max_read_len - c->K + 1);
	const size_t per_wave = ((size_t)max_kmers + 2 * MAX_HITS) * sizeof(uint64_t);
	int waves = 4;
	while (waves > 1 && per_wave * waves > 48 * 1024) waves >>= 1;
	if (per_wave > 64 * 1024)
		return fail(SDT_EINVAL, "reads of %llu bases do not fit the per-wavefront LDS window", (unsigned long long)max_read_len);
	if (!c->index_final) {
		int rcs = sync_stats(c);                          // counts stay readable through finish_count (host copy)
		if (rcs != SDT_OK) return rcs;
		const int g = scan_grid(c, view_slots(c));
		if (c->nw == 1) hipLaunchKernelGGL(k_finalize_contig_index<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_finalize_contig_index<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		else hipLaunchKernelGGL(k_finalize_contig_index<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), (const uint32_t *)c->d_ctg_ids, c->ctg_ord, c->d_stats);
		HIPCHK(hipGetLastError());
		c->index_final = true;
	}
	if (!c->d_hit_cursor) HIPCHK(hipMalloc((void **)&c->d_hit_cursor, sizeof(unsigned long long)));
	{
		const unsigned long long first_extra = nreads;      // hits[0 .. nreads) = first hit of each read, the rest follows
		HIPCHK(hipMemcpyAsync(c->d_hit_cursor, &first_extra, sizeof first_extra, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	uint64_t blocks = (nreads + waves - 1) / waves;
	const uint64_t cap = (uint64_t)c->cu_count * 32;
	if (blocks > cap) blocks = cap;
	if (blocks == 0) blocks = 1;
	EventPair *ev = next_event(c);
	if (ev) HIPCHK(hipEventRecord(ev->a, c->stream));
#define ALIGN_LAUNCH(NWV) hipLaunchKernelGGL(k_align_reads<NWV>, dim3((unsigned)blocks), dim3(TPB), per_wave * waves, c->stream, d_words, d_offs, nreads, \
	d_align_len, align_len_all, c->K, table_of<NWV>(c), (const uint32_t *)c->d_ctg_len, \
	(const uint32_t *)c->d_ctg_twin, c->num_ctg, max_kmers, waves, d_info, d_hits, (unsigned long long)max_hits, c->d_hit_cursor, c->d_stats)
	if (c->nw == 1) ALIGN_LAUNCH(1);
	else if (c->nw == 2) ALIGN_LAUNCH(2);
	else ALIGN_LAUNCH(4);
#undef ALIGN_LAUNCH
	HIPCHK(hipGetLastError());
	if (ev) {
		HIPCHK(hipEventRecord(ev->b, c->stream));
		ev->kmers = 0;
	}
	return SDT_OK;
}

int sdt_gpu_align_reads_device(sdt_ctx *c, const void *d_packed_words, const void *d_offsets, uint64_t nreads, uint64_t max_read_len,
                               const void *d_align_len, int align_len_all, void *d_read_info, void *d_hits, uint64_t max_hits,
                               uint64_t *nhits)
{
	if (!c || !d_packed_words || !d_offsets || !d_read_info || !d_hits)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_hits < nreads)
		return fail(SDT_EINVAL, "hits[] must hold at least one entry per read (%llu < %llu)", (unsigned long long)max_hits, (unsigned long long)nreads);
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX");
	HIPCHK(hipSetDevice(c->device));
	int rc = launch_align(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, nreads, max_read_len,
	                      (const int32_t *)d_align_len, align_len_all, (uint64_t *)d_read_info, (Hit *)d_hits, max_hits);
	if (rc != SDT_OK) return rc;
	unsigned long long h = 0;
	HIPCHK(hipMemcpyAsync(&h, c->d_hit_cursor, sizeof h, hipMemcpyDeviceToHost, c->stream));
	rc = sync_stats(c);
	if (rc != SDT_OK)
		return fail(SDT_ESTATE, "sdt_gpu_align_reads: %llu reads are longer than max_read_len or hit a contig outside the contig table",
		            (unsigned long long)c->h_stats->probe_fail);
	if (nhits) *nhits = h;
	if (h > max_hits)
		return fail(SDT_EFULL, "hit array holds %llu, the batch produced %llu", (unsigned long long)max_hits, h);
	return SDT_OK;
}

int sdt_gpu_align_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                        const int32_t *align_len, int align_len_all, uint64_t *read_info, sdt_hit *hits, uint64_t max_hits,
                        uint64_t *nhits)
{
	if (!c || !packed_words || !offsets || !read_info || (!hits && max_hits))
		return fail(SDT_EINVAL, "NULL argument");
	if (!(c->flags & SDT_FLAG_CONTIG_INDEX))
		return fail(SDT_ESTATE, "init with SDT_FLAG_CONTIG_INDEX");
	if (nreads == 0) { if (nhits) *nhits = 0; return SDT_OK; }
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)i);
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	}
	if (((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words", (unsigned long long)(((offsets[nreads] + 15) >> 4) + TAIL_PAD), TAIL_PAD);
	HIPCHK(hipSetDevice(c->device));
	int rc = ab_reserve(c, 0, nwords * sizeof(uint32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 1, (nreads + 1) * sizeof(uint64_t));
	if (rc == SDT_OK && align_len) rc = ab_reserve(c, 2, nreads * sizeof(int32_t));
	if (rc == SDT_OK) rc = ab_reserve(c, 3, nreads * sizeof(uint64_t));
	if (max_hits < nreads)
		return fail(SDT_EINVAL, "hits[] must hold at least one entry per read (%llu < %llu)", (unsigned long long)max_hits, (unsigned long long)nreads);
	if (rc == SDT_OK) rc = ab_reserve(c, 4, (max_hits ? max_hits : 1) * sizeof(Hit));
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemcpyAsync(c->ab[0], packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(c->ab[1], offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	if (align_len) HIPCHK(hipMemcpyAsync(c->ab[2], align_len, nreads * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	uint64_t got = 0;
	rc = sdt_gpu_align_reads_device(c, c->ab[0], c->ab[1], nreads, maxlen, align_len ? c->ab[2] : nullptr, align_len_all, c->ab[3], c->ab[4],
	                                max_hits, &got);
	if (nhits) *nhits = got;
	if (rc != SDT_OK) return rc;
	HIPCHK(hipMemcpyAsync(read_info, c->ab[3], nreads * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	if (got) HIPCHK(hipMemcpyAsync(hits, c->ab[4], got * sizeof(Hit), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return SDT_OK;
}
} // extern "C"
