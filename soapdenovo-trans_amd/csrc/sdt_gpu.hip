// sdt_gpu.hip -- kernels + C ABI (include/sdt_gpu.h) of the MI355X-native pregraph hashing path.
//
// gfx950 only.  No CPU fallback: every entry point needs a live HIP device.
//
// Kernels (all integer / HBM-bound, no MFMA):
//   k_count_reads<NW>     chopKmer4read (prlHashReads.c:164-310) fused with put_kmerset
//                         (newhash.c:411-462): a workgroup stages a tile of packed reads in LDS with
//                         coalesced loads, every lane cuts its k-mers out of LDS by funnel shift and
//                         updates the node table with one 64-bit atomic per occurrence.
//   k_sk_*                the locality pipeline (sdt_superkmer.cuh): minimizer buckets of super-k-mer records counted in LDS
//   k_delow<NW>           thread_delow   (prlHashReads.c:844-887)
//   k_mark_hist<NW>       thread_mark    (prlHashReads.c:911-967) + per-thread kmerFreq bins
//   k_export<NW>          compaction of the table into kmer_t-shaped arrays (inc/newhash.h:65-77)
//   k_rehash<NW>          table growth (the analogue of encap_kmerset, newhash.c:293-409)
#include "sdt_ctx.hpp"
#include "sdt_pipeline.hpp"
#include "sdt_table_kernels.cuh"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static std::atomic<int> g_live_ctx{0};                       // contexts alive in this process (the arena is trimmed when the last one goes)

int sdti::fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}
// slots of a flat table for `nodes` nodes: a load of SDT_TABLE_LOAD percent (default 45: measured best on the headline workload, between what the probes of the merges
// like and what the scans of the table cost), a multiple of 4096, 2^16 at least
uint64_t flat_slots_for(uint64_t nodes)
{
	static const int pct = sdt_tuning_env("SDT_TABLE_LOAD") && atoi(sdt_tuning_env("SDT_TABLE_LOAD")) >= 10 && atoi(sdt_tuning_env("SDT_TABLE_LOAD")) <= 69 ? atoi(sdt_tuning_env("SDT_TABLE_LOAD")) : 45;
	uint64_t slots = (uint64_t)((double)nodes * 100.0 / pct) + 4095;
	slots &= ~4095ULL;
	return slots < (1ULL << 16) ? (1ULL << 16) : slots;
}

void *keep_alloc(sdt_ctx *c, size_t bytes)
{
	bytes = (bytes + 255) & ~(size_t)255;
	if (c->keep_slabs.empty() || c->keep_slabs.back().size - c->keep_slabs.back().used < bytes) {
		sdt_ctx::KeepSlab sl;
		sl.size = bytes > ((size_t)1 << 30) ? bytes : ((size_t)1 << 30);
		sl.used = 0;
		sl.p = nullptr;
		if (hipMalloc((void **)&sl.p, sl.size) != hipSuccess) {
			sl.size = bytes;                             // a full slab does not fit any more: exactly what is needed
			if (hipMalloc((void **)&sl.p, sl.size) != hipSuccess)
				return nullptr;
		}
		c->keep_slabs.push_back(sl);
	}
	sdt_ctx::KeepSlab &b = c->keep_slabs.back();
	void *r = b.p + b.used;
	b.used += bytes;
	return r;
}

void keep_release(sdt_ctx *c)
{
	for (auto &sl : c->keep_slabs) (void)hipFree(sl.p);
	c->keep_slabs.clear();
	c->kept.clear();
}


int launch_clear(sdt_ctx *c, void *ent, uint32_t *aux, uint64_t *first, uint64_t slots)
{
	const int g = scan_grid(c, slots);
	if (c->nw == 1) { Table<1> t{(Entry<1> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<1>, dim3(g), dim3(TPB), 0, c->stream, t); }
	else if (c->nw == 2) { Table<2> t{(Entry<2> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<2>, dim3(g), dim3(TPB), 0, c->stream, t); }
	else { Table<4> t{(Entry<4> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_clear<4>, dim3(g), dim3(TPB), 0, c->stream, t); }
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

int alloc_table(sdt_ctx *c, uint64_t slots, void **ent, uint32_t **aux, uint64_t **first)
{
	*ent = nullptr;
	*aux = nullptr;
	*first = nullptr;
	hipError_t e = hipMalloc(ent, slots * entry_bytes(c->nw));
	if (e == hipSuccess)
		e = hipMalloc((void **)aux, slots * sizeof(uint32_t));
	if (e == hipSuccess && (c->flags & SDT_FLAG_TRACK_FIRST))
		e = hipMalloc((void **)first, slots * sizeof(uint64_t));
	if (e != hipSuccess) {
		if (*ent) (void)hipFree(*ent);
		if (*aux) (void)hipFree(*aux);
		*ent = nullptr;
		*aux = nullptr;
		return fail(SDT_ENOMEM, "node table: hipMalloc(%llu slots x %zu B) failed: %s", (unsigned long long)slots,
		            entry_bytes(c->nw) + 4, hipGetErrorString(e));
	}
	return SDT_OK;
}

static int drain_staged(sdt_ctx *c, bool force);

int sync_stats(sdt_ctx *c)
{
	// batches that were pushed but not launched yet, and work parked in the locality pipeline, belong to the table before
	// anybody looks at it
	if (!c->draining) {
		const int rcd = drain_staged(c, true);
		if (rcd != SDT_OK)
			return rcd;
	}
	if (c->sk.pending_kmers && !c->sk.flushing) {
		const int rcf = sk_flush(c);
		if (rcf != SDT_OK)
			return rcf;
	}
	HIPCHK(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream));
	{ const int rcw = c->comm.sync_watched(c->stream, "the drain of pass 1"); if (rcw != SDT_OK) return rcw; }
	if (c->h_stats->probe_fail)
		return fail(SDT_EFULL, "%llu inserts found no slot (table over-full or route bucket overflow)",
		            (unsigned long long)c->h_stats->probe_fail);
	// conservation in the locality pipeline: every k-mer cut into a record must come out of the count stage.  (A
	// workgroup geometry that lost a chunk now and then was found with exactly this comparison; it costs two counters.)
	if (!c->sk.flushing) {
		if (c->h_stats->sk_counted != c->sk.l2_in_total)
			return fail(SDT_ESTATE, "locality pipeline lost k-mers: %llu entered the count stage, %llu were counted",
			            (unsigned long long)c->sk.l2_in_total, (unsigned long long)c->h_stats->sk_counted);
		if (!c->sk.exchanged && c->h_stats->sk_emitted != c->sk.l2_in_total)
			return fail(SDT_ESTATE, "locality pipeline lost k-mers: %llu were cut into records, %llu reached the count stage",
			            (unsigned long long)c->h_stats->sk_emitted, (unsigned long long)c->sk.l2_in_total);
	}
	c->distinct_known = c->h_stats->distinct;
	c->kmers_known = c->h_stats->kmers;
	c->kmers_since_sync = 0;
	c->hard_since_sync = 0;
	return SDT_OK;
}

int grow_table(sdt_ctx *c, uint64_t need_nodes)
{
	// (any number of slots: what the nodes need at the load a fresh table is sized for, at least half as many again as before)
	uint64_t slots = flat_slots_for(need_nodes);
	if (slots < c->slots + c->slots / 2) slots = c->slots + c->slots / 2;
	void *ent = nullptr;
	uint32_t *aux = nullptr;
	uint64_t *first = nullptr;
	int rc = alloc_table(c, slots, &ent, &aux, &first);
	if (rc != SDT_OK && c->sk.ready && !c->sk.flushing) {
		// the locality pipeline's pools are only a cache of work: count what they hold, give the memory back, try again
		rc = sk_flush(c);
		if (rc == SDT_OK) {
			HIPCHK(hipStreamSynchronize(c->stream));
			sk_free(c);
			rc = alloc_table(c, slots, &ent, &aux, &first);
		}
	}
	if (rc != SDT_OK)
		return fail(SDT_EFULL, "cannot grow node table to %llu slots: %s", (unsigned long long)slots, g_err);
	rc = launch_clear(c, ent, aux, first, slots);
	if (rc != SDT_OK)
		return rc;
	const int g = scan_grid(c, c->slots);
	if (c->nw == 1) { Table<1> d{(Entry<1> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d, c->d_stats); }
	else if (c->nw == 2) { Table<2> d{(Entry<2> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d, c->d_stats); }
	else { Table<4> d{(Entry<4> *)ent, aux, slots, first}; hipLaunchKernelGGL(k_rehash<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d, c->d_stats); }
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipFree(c->d_ent));
	HIPCHK(hipFree(c->d_aux));
	if (c->d_first) HIPCHK(hipFree(c->d_first));
	c->d_ent = ent;
	c->d_aux = aux;
	c->d_first = first;
	c->slots = slots;
	return SDT_OK;
}

// make sure `incoming` more occurrences cannot push the table past MAX_LOAD (every occurrence might
// be a new node); syncs only when the cheap upper bound says it could
int ensure_room(sdt_ctx *c, uint64_t incoming)
{
	const double room = (double)c->slots * MAX_LOAD;
	if ((double)(c->distinct_known + c->kmers_since_sync + incoming) <= room)
		return SDT_OK;
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if ((double)(c->distinct_known + incoming) <= room)
		return SDT_OK;
	return grow_table(c, c->distinct_known + incoming);
}

EventPair *next_event(sdt_ctx *c)
{
	if (c->ev_used == c->ev.size()) {
		EventPair p;
		if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)
			return nullptr;
		p.kmers = 0;
		p.stage = 0;
		c->ev.push_back(p);
	}
	c->ev[c->ev_used].stage = 0;
	return &c->ev[c->ev_used++];
}

size_t tile_smem_bytes(int max_tile_words)
{
	return (size_t)(2 * (TILE_READS + 1) + 2 + LDS_LEAD + max_tile_words) * sizeof(uint32_t);
}

// the largest tile a batch can produce: TILE_READS consecutive reads; computed on the host side from the
// maximum read length the caller promised (offsets are device resident for the device entry point)
int tile_words_for(uint64_t max_read_len)
{
	const uint64_t bases = (uint64_t)TILE_READS * max_read_len + 16;
	return (int)((bases + 15) / 16) + TAIL_PAD + 1;
}
// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *sdt_gpu_last_error(void) { return g_err; }
int sdt_gpu_abi_version(void) { return SDT_ABI_VERSION; }

uint64_t sdt_owner_hash(const uint64_t *key_words_msw_first, int nwords)
{
	if (nwords == 1) { Key<1> k{{key_words_msw_first[0]}}; return key_hash<1>(k); }
	if (nwords == 2) { Key<2> k{{key_words_msw_first[0], key_words_msw_first[1]}}; return key_hash<2>(k); }
	Key<4> k{{key_words_msw_first[0], key_words_msw_first[1], key_words_msw_first[2], key_words_msw_first[3]}};
	return key_hash<4>(k);
}

int sdt_gpu_init(sdt_ctx **out, int device, int K, uint64_t est_distinct, uint32_t flags)
{
	if (!out)
		return fail(SDT_EINVAL, "ctx is NULL");
	*out = nullptr;
	if (K < 13 || K > 127 || (K & 1) == 0)
		return fail(SDT_EINVAL, "K must be odd and in 13..127 (got %d); apply call_pregraph's clamp first", K);
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(SDT_ENODEV, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
	if (device < 0 || device >= ndev)
		return fail(SDT_EINVAL, "device %d out of range (have %d)", device, ndev);
	{   // SDT_SYNC=block|yield|spin: how host threads wait for the device (a host that parses on every core it may use wants its
		// waiting thread off the CPU; the default is the runtime's own choice).  Must be set before the device is first used.
		const char *sm = sdt_tuning_env("SDT_SYNC");
		if (sm && *sm) {
			const unsigned fl = !strcmp(sm, "block") ? hipDeviceScheduleBlockingSync : !strcmp(sm, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleSpin;
			(void)hipSetDeviceFlags(fl);
			(void)hipGetLastError();
		}
	}
	e = hipSetDevice(device);
	if (e != hipSuccess)
		return fail(SDT_ENODEV, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess)
		return fail(SDT_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(SDT_ENODEV, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

	sdt_ctx *c = new (std::nothrow) sdt_ctx();
	if (!c)
		return fail(SDT_ENOMEM, "out of host memory");
	g_live_ctx.fetch_add(1);
	c->device = device;
	c->K = K;
	c->nw = K <= 31 ? 1 : (K <= 63 ? 2 : 4);
	c->flags = (flags & SDT_FLAG_CONTIG_INDEX) ? (flags | SDT_FLAG_TRACK_FIRST) : flags;   // the index lives in the first-occurrence slot
	c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (est_distinct == 0)
		est_distinct = 1ULL << 22;
	// one node table for both kernel families (the direct one counts into it with one atomic per occurrence, the locality pipeline
	// merges its LDS tables into it), sized by the caller's estimate and grown by k_rehash.  (Round 5 also built a node LOG folded
	// into a bucket-major table; it ran at the speed of these merges and was removed in round 6: DESIGN.md section 3.)
	c->slots = flat_slots_for(est_distinct);
#define INIT_CHK(expr)                                                                    \
	do {                                                                                  \
		hipError_t e2_ = (expr);                                                          \
		if (e2_ != hipSuccess) {                                                          \
			int rc_ = fail(e2_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e2_)); \
			sdt_gpu_destroy(c);                                                           \
			return rc_;                                                                   \
		}                                                                                 \
	} while (0)
	INIT_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	INIT_CHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
	for (int i = 0; i < sdt_ctx::NSTAGE; i++) {
		INIT_CHK(hipEventCreateWithFlags(&c->buf_free[i], hipEventDisableTiming));
		INIT_CHK(hipEventCreateWithFlags(&c->copied[i], hipEventDisableTiming));
	}
	INIT_CHK(hipMalloc((void **)&c->d_stats, sizeof(Stats)));
	INIT_CHK(hipHostMalloc((void **)&c->h_stats, sizeof(Stats), hipHostMallocDefault));
	INIT_CHK(hipMalloc((void **)&c->d_hist, 257 * sizeof(unsigned long long)));
	int rc = alloc_table(c, c->slots, &c->d_ent, &c->d_aux, &c->d_first);
	if (rc != SDT_OK) {
		sdt_gpu_destroy(c);
		return rc;
	}
	rc = sdt_gpu_reset(c);
	if (rc != SDT_OK) {
		sdt_gpu_destroy(c);
		return rc;
	}
	*out = c;
	return SDT_OK;
}

int sdt_gpu_destroy(sdt_ctx *c)
{
	if (!c)
		return SDT_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
	for (auto &p : c->ev) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
	for (int i = 0; i < sdt_ctx::NSTAGE; i++) {
		if (c->d_words[i]) (void)hipFree(c->d_words[i]);
		if (c->d_offs[i]) (void)hipFree(c->d_offs[i]);
		if (c->buf_free[i]) (void)hipEventDestroy(c->buf_free[i]);
		if (c->copied[i]) (void)hipEventDestroy(c->copied[i]);
	}
	if (c->d_ent) (void)hipFree(c->d_ent);
	if (c->d_aux) (void)hipFree(c->d_aux);
	if (c->d_first) (void)hipFree(c->d_first);
	if (c->d_stats) (void)hipFree(c->d_stats);
	if (c->h_stats) (void)hipHostFree(c->h_stats);
	if (c->d_hist) (void)hipFree(c->d_hist);
	keep_release(c);
	if (c->d_patch) (void)hipFree(c->d_patch);
	if (c->d_arcs) (void)hipFree(c->d_arcs);
	if (c->d_idx) (void)hipFree(c->d_idx);
	if (c->gx) sdti::graph_ext_free(c->gx);
	if (c->d_ctg_ids) (void)hipFree(c->d_ctg_ids);
	if (c->d_ctg_len) (void)hipFree(c->d_ctg_len);
	if (c->d_ctg_twin) (void)hipFree(c->d_ctg_twin);
	if (c->d_hit_cursor) (void)hipFree(c->d_hit_cursor);
	for (int i = 0; i < 5; i++) if (c->ab[i]) (void)hipFree(c->ab[i]);
	sk_free(c);
	shard_free(c);
	c->comm.close_all();
	if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
	if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
	delete c;
	if (g_live_ctx.fetch_sub(1) == 1) (void)sdti::mem_trim();       // the last context of the process: the arena's memory goes back to the driver
	return SDT_OK;
}

int sdt_gpu_reset(sdt_ctx *c)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	if ((c->flags & SDT_FLAG_TRACK_FIRST) && !c->d_first)           // (dropped once the device had laid the graph out: sdti::drop_first)
		HIPCHK(hipMalloc((void **)&c->d_first, c->slots * sizeof(uint64_t)));
	int rc = launch_clear(c, c->d_ent, c->d_aux, c->d_first, c->slots);
	if (rc != SDT_OK)
		return rc;
	HIPCHK(hipMemsetAsync(c->d_stats, 0, sizeof(Stats), c->stream));
	c->sk.l2_in_total = 0;
	c->sk.stream_flushes = 0;
	c->sk.exchanged = false;
	c->distinct_known = 0;
	c->kmers_known = 0;
	c->kmers_since_sync = 0;
	c->hard_since_sync = 0;
	c->kmers_total_host = 0;
	c->kmers_offered = 0;
	c->expect_kmers = 0;
	c->ord_base = 0;
	c->push_ord_base = 0;
	c->push_ord_stride = 1;
	c->staged.clear();
	c->staged_head = 0;
	c->ord_stride = 1;
	if (c->sk.ready) {                               // records scattered but not counted belong to the run being forgotten
		const int rcr = sk_reset_pool1(c);
		if (rcr != SDT_OK)
			return rcr;
	}
	keep_release(c);
	c->ctg_ord = 0;
	c->index_final = false;
	c->paths_loaded = false;
	c->sh.pending = false;                           // (a sharded call that failed half-way leaves nothing behind)
	c->sh.items.clear();
	if (c->d_idx) (void)hipFree(c->d_idx);
	c->d_idx = nullptr;
	c->idx_slots = c->idx_n = 0;
	return SDT_OK;
}

// pinned host memory for batches that are pushed asynchronously (the DMA engine reads it directly; pageable memory goes through the
// runtime's own staging copy at a fraction of the link).  Any host thread may call these.
void *sdt_gpu_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	return p;
}
void sdt_gpu_host_free(void *p)
{
	if (p) (void)hipHostFree(p);
}

int sdt_gpu_key_words(const sdt_ctx *c) { return c ? c->nw : 0; }
uint64_t sdt_gpu_table_slots(const sdt_ctx *c) { return c ? view_slots(c) : 0; }
void *sdt_gpu_stream(const sdt_ctx *c) { return c ? (void *)c->stream : nullptr; }
int sdt_gpu_set_read_ordinal(sdt_ctx *c, uint64_t base, uint64_t stride)
{
	if (!c || stride == 0)
		return fail(SDT_EINVAL, "bad argument");
	c->push_ord_base = base;
	c->push_ord_stride = stride;
	if (c->staged_head == c->staged.size()) {            // nothing queued: the launch cursor follows at once
		c->ord_base = base;
		c->ord_stride = stride;
	}
	return SDT_OK;
}

// run the kernels on a stream the caller owns (e.g. the stream a collective library orders against)
int sdt_gpu_set_stream(sdt_ctx *c, void *hip_stream)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->own_stream) {
		HIPCHK(hipStreamDestroy(c->stream));
		c->own_stream = false;
	}
	if (hip_stream) {
		c->stream = (hipStream_t)hip_stream;
	} else {
		HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->own_stream = true;
	}
	return SDT_OK;
}
// launch the fused chop+insert kernel on a device-resident batch, in chunks of reads small enough that
// "every occurrence is a new node" cannot overflow the table between two looks at the node counter.
static const uint64_t CHUNK_KMERS = 1ULL << 27;



static int launch_count(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads,
                        uint64_t max_read_len)
{
	if (nreads == 0)
		return SDT_OK;
	if (max_read_len < (uint64_t)c->K + 1)
		return SDT_OK;                               // no read can hold a k-mer (prlHashReads.c:592)
	const int mtw = tile_words_for(max_read_len);
	const size_t smem = tile_smem_bytes(mtw);
	if (smem > 64 * 1024)
		return fail(SDT_EINVAL, "max read length %llu needs %zu B of LDS per tile (limit 64 KiB)",
		            (unsigned long long)max_read_len, smem);
	const uint64_t per_read = max_read_len - c->K + 1;
	// The locality pipeline is opt-in in round 1: on MI355X it measures 11 G k-mers/s against the direct
	// kernel's 19.5 G (profiles/r1/partition_pipeline_50M.md has the per-stage rates and what has to change).
	// default: the locality pipeline wherever it applies (2.2x the direct kernel on 200 M x 150 bp, K = 31); SDT_FLAG_DIRECT /
	// SDT_FLAG_PARTITION force one family (the latter still needs a geometry the pipeline can take)
	// (a small job is not worth the pipeline's fixed cost -- two host syncs and scans over 2^18 buckets, ~3 ms -- unless asked for)
	const bool sk_small = !(c->flags & SDT_FLAG_PARTITION) && !c->sk.pending_kmers && c->kmers_offered + nreads * per_read < (1ULL << 27) &&
	                      c->expect_kmers < (1ULL << 27);      // (a caller that knows more is coming says so: sdt_gpu_hint_total_kmers)
	c->kmers_offered += nreads * per_read;
	const bool sk_ord_ok = c->ord_base + nreads * c->ord_stride < SK_MAX_READ_ORDINAL;      // what a record header can number
	if (!(c->flags & SDT_FLAG_DIRECT) && !sk_small && sk_ord_ok && sk_applicable(c, max_read_len)) {
		const int rcs = sk_scatter(c, d_words, d_offs, nreads, max_read_len);
		if (rcs == SDT_OK)
			c->ord_base += nreads * c->ord_stride;     // the next batch continues the read stream
		return rcs;
	}
	uint64_t chunk_reads = CHUNK_KMERS / per_read;
	chunk_reads = chunk_reads / TILE_READS * TILE_READS;
	if (chunk_reads < TILE_READS)
		chunk_reads = TILE_READS;
	for (uint64_t r0 = 0; r0 < nreads; r0 += chunk_reads) {
		const uint64_t nr = nreads - r0 < chunk_reads ? nreads - r0 : chunk_reads;
		const uint64_t upper = nr * per_read;
		int rc = ensure_room(c, upper);
		if (rc != SDT_OK)
			return rc;
		const uint64_t ntiles = (nr + TILE_READS - 1) / TILE_READS;
		uint64_t grid = ntiles;
		const uint64_t cap = (uint64_t)c->cu_count * 8;
		if (grid > cap) grid = cap;
		EventPair *ev = next_event(c);
		if (!ev)
			return fail(SDT_EHIP, "hipEventCreate failed");
		ev->kmers = upper;
		HIPCHK(hipEventRecord(ev->a, c->stream));
		// offsets are absolute base indices into d_words, so a sub-range of reads is just a shifted pointer
		if (c->nw == 1)
			hipLaunchKernelGGL(k_count_reads<1>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<1>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		else if (c->nw == 2)
			hipLaunchKernelGGL(k_count_reads<2>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<2>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		else
			hipLaunchKernelGGL(k_count_reads<4>, dim3((unsigned)grid), dim3(TPB), smem, c->stream, d_words, d_offs + r0, nr, c->K, mtw, flat_of<4>(c), c->d_stats, c->ord_base + r0 * c->ord_stride, c->ord_stride);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(ev->b, c->stream));
		c->kmers_since_sync += upper;
	}
	c->ord_base += nreads * c->ord_stride;         // the next batch continues the read stream
	return SDT_OK;
}

// enqueue one host batch: H2D on the copy stream into the next buffer of the ring, kernels behind it.  Returns without waiting
// for the copy; *ticket (may be NULL) names it for sdt_gpu_push_wait.
// offsets of a batch of equal-length reads, made where they are used (8 B per read that need not cross PCIe: 21 % of a 150-bp batch)
__global__ __launch_bounds__(256) void k_fixed_offsets(uint64_t *offs, uint64_t nreads, uint64_t len)
{
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= nreads; i += (uint64_t)gridDim.x * 256ull)
		offs[i] = i * len;
}

static int push_reads_enqueue(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                              uint64_t *ticket, uint64_t fixed_len = 0)
{
	if (!c || (!packed_words && nwords) || (!offsets && !fixed_len))
		return fail(SDT_EINVAL, "NULL argument");
	if (ticket) *ticket = c->push_ticket;
	if (nreads == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	// nothing queued: the launch cursor is the truth (calls that count device-resident reads advance it on their own, and
	// sdt_gpu_set_read_ordinal sets both when the queue is empty)
	if (c->staged_head == c->staged.size()) {
		c->push_ord_base = c->ord_base;
		c->push_ord_stride = c->ord_stride;
	}
	// batch geometry from the host copy of the offsets
	uint64_t kmers = 0, maxlen = 0, bad = 0;
	const uint64_t Kp1 = (uint64_t)c->K + 1;
	if (fixed_len) {
		maxlen = fixed_len;
		kmers = fixed_len >= Kp1 ? nreads * (fixed_len - Kp1 + 2) : 0;
	} else {
		for (uint64_t i = 0; i < nreads; i++) {          // (branch-free: one pass over a million offsets per batch, vectorised)
			const uint64_t len = offsets[i + 1] - offsets[i];
			bad |= len >> 63;                            // offsets[i + 1] < offsets[i]
			maxlen = len > maxlen ? len : maxlen;
			kmers += len >= Kp1 ? len - Kp1 + 2 : 0;
		}
	}
	if (bad)
		return fail(SDT_EINVAL, "offsets not monotonic");
	const uint64_t total_bases = fixed_len ? nreads * fixed_len : offsets[nreads];
	if (((total_bases + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short: need %llu words incl. %d pad words, got %llu",
		            (unsigned long long)(((total_bases + 15) >> 4) + TAIL_PAD), TAIL_PAD, (unsigned long long)nwords);
	const int b = c->next_buf;
	uint32_t *dw;
	uint64_t *dof;
	if (c->flags & SDT_FLAG_KEEP_READS) {
		// the batch stays resident for the second pass: own buffers instead of the recycled staging ring
		sdt_ctx::KeptBatch kb;
		kb.nwords = nwords; kb.nreads = nreads; kb.ord_base = c->push_ord_base; kb.ord_stride = c->push_ord_stride; kb.maxlen = maxlen;
		kb.d_words = (uint32_t *)keep_alloc(c, nwords * sizeof(uint32_t));
		kb.d_offs = (uint64_t *)keep_alloc(c, (nreads + 1) * sizeof(uint64_t));
		if (!kb.d_words || !kb.d_offs)
			return fail(SDT_ENOMEM, "kept reads: no device memory for another batch (%zu slabs held); run with --host-map", c->keep_slabs.size());
		c->kept.push_back(kb);
		dw = kb.d_words;
		dof = kb.d_offs;
	} else {
		// the kernels that last read this staging buffer must be done before it is overwritten (NSTAGE batches ago): its batch
		// has left the queue (drain_staged never lets the queue grow to NSTAGE) and its buf_free event is the newest one
		if (c->staged.size() - c->staged_head + 1 >= (size_t)sdt_ctx::NSTAGE) {
			const int rcd = drain_staged(c, true);
			if (rcd != SDT_OK) return rcd;
		}
		HIPCHK(hipEventSynchronize(c->buf_free[b]));
		// (capacities with head room: the chunks of a file differ by a few words, and every batch that sets a new record would
		// otherwise cost a hipFree -- a device-wide synchronisation -- and a hipMalloc: 200 M reads in 32-MiB chunks spent
		// seconds there)
		if (c->cap_words[b] < nwords) {
			if (c->d_words[b]) HIPCHK(hipFree(c->d_words[b]));
			c->d_words[b] = nullptr;
			c->cap_words[b] = 0;
			const uint64_t cap = nwords + nwords / 8 + (1u << 16);
			HIPCHK(hipMalloc((void **)&c->d_words[b], cap * sizeof(uint32_t)));
			c->cap_words[b] = cap;
		}
		if (c->cap_offs[b] < nreads + 1) {
			if (c->d_offs[b]) HIPCHK(hipFree(c->d_offs[b]));
			c->d_offs[b] = nullptr;
			c->cap_offs[b] = 0;
			const uint64_t cap = nreads + 1 + nreads / 8 + (1u << 12);
			HIPCHK(hipMalloc((void **)&c->d_offs[b], cap * sizeof(uint64_t)));
			c->cap_offs[b] = cap;
		}
		dw = c->d_words[b];
		dof = c->d_offs[b];
	}
	HIPCHK(hipMemcpyAsync(dw, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->copy_stream));
	if (!fixed_len)
		HIPCHK(hipMemcpyAsync(dof, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream));
	HIPCHK(hipEventRecord(c->copied[b], c->copy_stream));
	if (c->staged_head == c->staged.size()) {
		c->staged.clear();
		c->staged_head = 0;
	}
	c->staged.push_back(sdt_ctx::Staged{dw, dof, nreads, maxlen, c->push_ord_base, c->push_ord_stride, b, fixed_len});
	c->push_ord_base += nreads * c->push_ord_stride;     // the next batch continues the read stream
	c->kmers_total_host += kmers;
	c->next_buf = (b + 1) % sdt_ctx::NSTAGE;
	c->push_ticket++;
	if (ticket) *ticket = c->push_ticket;
	return drain_staged(c, false);
}

// would launching this batch make the locality pipeline flush (= block the host)?  (an estimate: sk_scatter decides)
static bool launch_would_flush(const sdt_ctx *c, const sdt_ctx::Staged &b)
{
	const sdt_ctx::SkState &k = c->sk;
	if (!k.ready || (c->flags & SDT_FLAG_DIRECT) || b.maxlen < (uint64_t)c->K + 1)
		return false;
	const uint64_t per_read = b.maxlen - c->K + 1;
	return k.pending_kmers + (b.nreads + SK_TILE_READS) * per_read > sk_batch_limit(c);
}

// launch the kernels of queued batches, oldest first; a launch that would flush waits for STAGE_AHEAD queued copies unless
// `force` (a sync point) or the ring is about to run out of slots
static int drain_staged(sdt_ctx *c, bool force)
{
	if (c->draining)
		return SDT_OK;
	c->draining = true;
	int rc = SDT_OK;
	bool launched = false;
	while (rc == SDT_OK && c->staged_head < c->staged.size()) {
		const sdt_ctx::Staged b = c->staged[c->staged_head];
		const size_t queued = c->staged.size() - c->staged_head;
		if (!force && queued < (size_t)sdt_ctx::STAGE_AHEAD && queued + 2 < (size_t)sdt_ctx::NSTAGE && launch_would_flush(c, b))
			break;
		c->staged_head++;
		launched = true;
		hipError_t e = hipStreamWaitEvent(c->stream, c->copied[b.slot], 0);
		if (e != hipSuccess) { rc = fail(SDT_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; }
		c->ord_base = b.ord_base;
		c->ord_stride = b.ord_stride;
		if (b.fixed_len) {
			hipLaunchKernelGGL(k_fixed_offsets, dim3(scan_grid(c, b.nreads + 1)), dim3(256), 0, c->stream, b.dof, b.nreads, b.fixed_len);
			e = hipGetLastError();
			if (e != hipSuccess) { rc = fail(SDT_EHIP, "k_fixed_offsets: %s", hipGetErrorString(e)); break; }
		}
		rc = launch_count(c, b.dw, b.dof, b.nreads, b.maxlen);
		if (rc != SDT_OK) break;
		e = hipEventRecord(c->buf_free[b.slot], c->stream);
		if (e != hipSuccess) { rc = fail(SDT_EHIP, "hipEventRecord: %s", hipGetErrorString(e)); break; }
	}
	if (launched && c->staged_head == c->staged.size()) {     // the launch cursor has caught up with the push cursor
		c->ord_base = c->push_ord_base;
		c->ord_stride = c->push_ord_stride;
	}
	if (!launched && c->staged_head == c->staged.size()) {    // nothing was queued: calls that count device-resident reads move the
		c->push_ord_base = c->ord_base;                       // launch cursor on their own, and the push cursor follows it
		c->push_ord_stride = c->ord_stride;
	}
	c->draining = false;
	return rc;
}

int sdt_gpu_push_reads_async(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                             uint64_t *ticket)
{
	return push_reads_enqueue(c, packed_words, nwords, offsets, nreads, ticket);
}

int sdt_gpu_push_reads_fixed_async(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, uint64_t nreads, uint64_t read_len,
                                   uint64_t *ticket)
{
	if (read_len == 0)
		return fail(SDT_EINVAL, "read_len must be > 0");
	return push_reads_enqueue(c, packed_words, nwords, nullptr, nreads, ticket, read_len);
}

int sdt_gpu_push_wait(sdt_ctx *c, uint64_t ticket)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (ticket == 0 || ticket > c->push_ticket)
		return ticket == 0 ? SDT_OK : fail(SDT_EINVAL, "ticket %llu was never issued", (unsigned long long)ticket);
	HIPCHK(hipSetDevice(c->device));
	// (copies run in order on one stream: should the ring slot have been reused since, its event stands for a LATER copy)
	HIPCHK(hipEventSynchronize(c->copied[(ticket - 1) % sdt_ctx::NSTAGE]));
	return SDT_OK;
}

int sdt_gpu_hint_total_kmers(sdt_ctx *c, uint64_t kmers)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	c->expect_kmers = kmers;
	// The node table for the job NOW -- while nothing is in it (growing it moves nothing) and BEFORE the pools of the locality pipeline
	// take their share of what is free: pools sized for the announced job on a device whose table then doubles several times left the
	// table's rebuilds fighting the pools for memory (200 M reads: count stage 0.3 -> 1-3 s, and every later table scan crawled).
	// One distinct node per 28 k-mers is what deep transcriptome data gives (C3: 35); data that repeats less grows the table as before.
	if (kmers && c->distinct_known == 0 && c->kmers_since_sync == 0 && c->kmers_total_host == 0 && !c->sk.ready && c->staged_head == c->staged.size()) {
		HIPCHK(hipSetDevice(c->device));
		size_t free_b = 0, total_b = 0;
		HIPCHK(sdti::mem_info(&free_b, &total_b));
		uint64_t est = kmers / 28;
		const uint64_t per_slot = entry_bytes(c->nw) + 4 + ((c->flags & SDT_FLAG_TRACK_FIRST) ? 8 : 0);
		while (est > (1u << 20) && (double)flat_slots_for(est) * (double)per_slot > (double)free_b * 0.3)
			est /= 2;                                    // (never more than ~30 % of what is free)
		if ((double)est > (double)c->slots * MAX_LOAD) {
			const int rc = grow_table(c, est);
			if (rc != SDT_OK) return rc;
		}
	}
	return SDT_OK;
}

int sdt_gpu_push_reads(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets,
                       uint64_t nreads)
{
	uint64_t t = 0;
	const int rc = push_reads_enqueue(c, packed_words, nwords, offsets, nreads, &t);
	if (rc != SDT_OK || nreads == 0)
		return rc;
	return sdt_gpu_push_wait(c, t);                  // the caller may reuse its buffers once the H2D copies have left them
}

int sdt_gpu_count_reads_device(sdt_ctx *c, const void *d_packed_words, uint64_t nwords, const void *d_offsets,
                               uint64_t nreads, uint64_t max_read_len)
{
	(void)nwords;
	if (!c || !d_packed_words || !d_offsets)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_read_len == 0)
		return fail(SDT_EINVAL, "max_read_len must be > 0");
	HIPCHK(hipSetDevice(c->device));
	const int rcd = drain_staged(c, true);               // (batches pushed earlier come first)
	if (rcd != SDT_OK) return rcd;
	return launch_count(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, nreads, max_read_len);
}

int sdt_gpu_finish_count(sdt_ctx *c, uint64_t *kmers_processed, uint64_t *nodes)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (kmers_processed) *kmers_processed = c->h_stats->kmers;
	if (nodes) *nodes = c->h_stats->distinct;
	return SDT_OK;
}

int sdt_gpu_delow(sdt_ctx *c, int d, uint64_t *removed)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	if (d < 0)
		d = 0;       // pregraph.c:159: negative -d becomes 0
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_delow<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), (uint32_t)d, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_delow<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), (uint32_t)d, c->d_stats);
	else hipLaunchKernelGGL(k_delow<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), (uint32_t)d, c->d_stats);
	HIPCHK(hipGetLastError());
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (removed) *removed = c->h_stats->scratch;
	return SDT_OK;
}

int sdt_gpu_mark_and_hist(sdt_ctx *c, int64_t hist[257], uint64_t *linear)
{
	if (!c || !hist)
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_staged(c, true); if (rcd != SDT_OK) return rcd; }
	HIPCHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	HIPCHK(hipMemsetAsync(c->d_hist, 0, 257 * sizeof(unsigned long long), c->stream));
	const int g = scan_grid(c, view_slots(c));
	if (c->nw == 1) hipLaunchKernelGGL(k_mark_hist<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), c->d_hist, c->d_stats);
	else if (c->nw == 2) hipLaunchKernelGGL(k_mark_hist<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), c->d_hist, c->d_stats);
	else hipLaunchKernelGGL(k_mark_hist<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), c->d_hist, c->d_stats);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(hist, c->d_hist, 257 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	if (linear) *linear = c->h_stats->scratch;
	return SDT_OK;
}

int sdt_gpu_export_nodes(sdt_ctx *c, uint64_t *keys, uint32_t *l_links, uint32_t *r_flags, uint32_t *count,
                         uint64_t *first, uint64_t max_nodes, uint64_t *n)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK)
		return rc;
	const uint64_t nodes = c->h_stats->distinct;
	if (n) *n = nodes;
	if (!keys && !l_links && !r_flags && !count && !first)
		return SDT_OK;
	if (first && !c->d_first)
		return fail(SDT_ESTATE, "first-occurrence ordinals were not tracked: init with SDT_FLAG_TRACK_FIRST");
	if (max_nodes < nodes)
		return fail(SDT_EINVAL, "export arrays hold %llu nodes, table has %llu", (unsigned long long)max_nodes,
		            (unsigned long long)nodes);
	uint64_t *d_keys = nullptr;
	uint32_t *d_l = nullptr, *d_r = nullptr, *d_c = nullptr;
	uint64_t *d_f = nullptr;
	const uint64_t m = nodes ? nodes : 1;
	int ret = SDT_OK;
#define EXP_CHK(expr)                                                                                  \
	do {                                                                                               \
		hipError_t e3_ = (expr);                                                                       \
		if (e3_ != hipSuccess) {                                                                       \
			ret = fail(e3_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e3_)); \
			goto done;                                                                                 \
		}                                                                                              \
	} while (0)
	if (keys) EXP_CHK(hipMalloc((void **)&d_keys, m * c->nw * sizeof(uint64_t)));
	if (l_links) EXP_CHK(hipMalloc((void **)&d_l, m * sizeof(uint32_t)));
	if (r_flags) EXP_CHK(hipMalloc((void **)&d_r, m * sizeof(uint32_t)));
	if (count) EXP_CHK(hipMalloc((void **)&d_c, m * sizeof(uint32_t)));
	if (first) EXP_CHK(hipMalloc((void **)&d_f, m * sizeof(uint64_t)));
	EXP_CHK(hipMemsetAsync(&c->d_stats->scratch, 0, sizeof(unsigned long long), c->stream));
	{
		const int g = scan_grid(c, view_slots(c));
		if (c->nw == 1) hipLaunchKernelGGL(k_export<1>, dim3(g), dim3(TPB), 0, c->stream, table_of<1>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_export<2>, dim3(g), dim3(TPB), 0, c->stream, table_of<2>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
		else hipLaunchKernelGGL(k_export<4>, dim3(g), dim3(TPB), 0, c->stream, table_of<4>(c), d_keys, d_l, d_r, d_c, d_f, (unsigned long long)nodes, c->d_stats);
	}
	EXP_CHK(hipGetLastError());
	if (keys) EXP_CHK(hipMemcpyAsync(keys, d_keys, nodes * c->nw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	if (l_links) EXP_CHK(hipMemcpyAsync(l_links, d_l, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (r_flags) EXP_CHK(hipMemcpyAsync(r_flags, d_r, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (count) EXP_CHK(hipMemcpyAsync(count, d_c, nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	if (first) EXP_CHK(hipMemcpyAsync(first, d_f, nodes * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	EXP_CHK(hipStreamSynchronize(c->stream));
done:
	if (d_keys) (void)hipFree(d_keys);
	if (d_l) (void)hipFree(d_l);
	if (d_r) (void)hipFree(d_r);
	if (d_c) (void)hipFree(d_c);
	if (d_f) (void)hipFree(d_f);
	return ret;
}
int sdt_gpu_release_table(sdt_ctx *c)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	int rc = sdti::release_pass1(c);                 // drains pass 1; the pools go back
	if (rc != SDT_OK) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	if (c->d_ent) (void)hipFree(c->d_ent);
	if (c->d_aux) (void)hipFree(c->d_aux);
	if (c->d_first) (void)hipFree(c->d_first);
	if (c->d_idx) { (void)hipFree(c->d_idx); c->d_idx = nullptr; c->idx_slots = c->idx_n = 0; }
	c->d_ent = nullptr; c->d_aux = nullptr; c->d_first = nullptr;
	c->slots = 0;
	HIPCHK(hipMemsetAsync(&c->d_stats->distinct, 0, sizeof(unsigned long long), c->stream));
	c->distinct_known = 0;
	c->kmers_since_sync = c->hard_since_sync = 0;
	return SDT_OK;
}
int sdt_gpu_kernel_time(sdt_ctx *c, int reset, double *ms, uint64_t *launches, uint64_t *kmers)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	double total = 0;
	uint64_t km = 0;
	for (size_t i = 0; i < c->ev_used; i++) {
		float t = 0;
		HIPCHK(hipEventElapsedTime(&t, c->ev[i].a, c->ev[i].b));
		total += t;
		km += c->ev[i].kmers;
	}
	if (ms) *ms = total;
	if (launches) *launches = c->ev_used;
	if (kmers) *kmers = km;
	if (reset)
		c->ev_used = 0;
	return SDT_OK;
}
int sdt_gpu_table_info(sdt_ctx *c, uint64_t info[8])
{
	if (!c || !info)
		return fail(SDT_EINVAL, "NULL argument");
	for (int i = 0; i < 8; i++) info[i] = 0;
	info[1] = view_slots(c);
	info[2] = c->distinct_known;
	return SDT_OK;
}
int sdt_gpu_import_nodes(sdt_ctx *c, const uint64_t *keys, const uint32_t *l_links, const uint32_t *r_flags, const uint32_t *count,
                         const uint64_t *first, uint64_t n)
{
	if (!c || (n && (!keys || !l_links || !r_flags || !count)))
		return fail(SDT_EINVAL, "NULL argument");
	if (n == 0)
		return SDT_OK;
	HIPCHK(hipSetDevice(c->device));
	int rc = sync_stats(c);
	if (rc != SDT_OK) return rc;
	if ((double)(c->distinct_known + n) > (double)c->slots * MAX_LOAD) {
		rc = grow_table(c, c->distinct_known + n);
		if (rc != SDT_OK) return rc;
	}
	uint64_t *d_k = nullptr, *d_f = nullptr;
	uint32_t *d_l = nullptr, *d_r = nullptr, *d_c = nullptr;
	const uint64_t STEP = 1ULL << 26;                // nodes per upload: bounded staging memory
	const uint64_t m = n < STEP ? n : STEP;
	HIPCHK(hipMalloc((void **)&d_k, m * c->nw * 8));
	HIPCHK(hipMalloc((void **)&d_l, m * 4));
	HIPCHK(hipMalloc((void **)&d_r, m * 4));
	HIPCHK(hipMalloc((void **)&d_c, m * 4));
	if (first && c->d_first) HIPCHK(hipMalloc((void **)&d_f, m * 8));
	for (uint64_t i0 = 0; i0 < n && rc == SDT_OK; i0 += STEP) {
		const uint64_t k = n - i0 < STEP ? n - i0 : STEP;
		HIPCHK(hipMemcpyAsync(d_k, keys + i0 * c->nw, k * c->nw * 8, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_l, l_links + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_r, r_flags + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_c, count + i0, k * 4, hipMemcpyHostToDevice, c->stream));
		if (d_f) HIPCHK(hipMemcpyAsync(d_f, first + i0, k * 8, hipMemcpyHostToDevice, c->stream));
		const int g = scan_grid(c, k);
		if (c->nw == 1) hipLaunchKernelGGL(k_import<1>, dim3(g), dim3(TPB), 0, c->stream, flat_of<1>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_import<2>, dim3(g), dim3(TPB), 0, c->stream, flat_of<2>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		else hipLaunchKernelGGL(k_import<4>, dim3(g), dim3(TPB), 0, c->stream, flat_of<4>(c), d_k, d_l, d_r, d_c, d_f, k, c->d_stats);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	(void)hipFree(d_k); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_c);
	if (d_f) (void)hipFree(d_f);
	return sync_stats(c);                            // a key that was already there shows up as SDT_EFULL
}
int sdt_gpu_stage_times(sdt_ctx *c, double ms[SDT_NSTAGES], uint64_t counters[SDT_NCOUNTERS])
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipStreamSynchronize(c->stream));
	if (ms) {
		for (int i = 0; i < SDT_NSTAGES; i++) ms[i] = 0;
		for (size_t i = 0; i < c->ev_used; i++) {
			float t = 0;
			HIPCHK(hipEventElapsedTime(&t, c->ev[i].a, c->ev[i].b));
			ms[c->ev[i].stage] += t;
		}
	}
	if (counters) {
		HIPCHK(hipMemcpy(c->h_stats, c->d_stats, sizeof(Stats), hipMemcpyDeviceToHost));
		counters[0] = c->h_stats->sk_merges;
		counters[1] = c->h_stats->sk_spills;
		counters[2] = c->h_stats->sk_direct;
		counters[3] = c->h_stats->sk_gens;
		counters[4] = c->sk.st_chunks1;
		counters[5] = c->sk.st_chunks2;
		counters[6] = c->sk.st_flushes;
		counters[7] = c->sk.cap_kmers;
		for (int i = 0; i < 4; i++)
			counters[8 + i] = c->h_stats->sk_cyc[i];
		for (int i = 0; i < 4; i++)
			counters[12 + i] = c->h_stats->sk_cyc1[i];
		counters[16] = c->h_stats->sk_distinct_recs;
		counters[17] = c->h_stats->sk_records;
		counters[18] = c->h_stats->sk_distinct_kmers;
		counters[19] = 0;
	}
	return SDT_OK;
}
} // extern "C"

int drain_pushes(sdt_ctx *c, bool force) { return drain_staged(c, force); }

// ---- what the graph unit (sdt_gpu_graph.hip) sees of a context ----------------------------------------------------
sdti::GraphView sdti::graph_view(sdt_ctx *c)
{
	GraphView v;
	v.device = c->device; v.K = c->K; v.nw = c->nw; v.cu_count = c->cu_count;
	v.slots = view_slots(c);
	v.d_ent = c->d_ent; v.d_aux = c->d_aux; v.d_first = c->d_first;
	v.d_stats = c->d_stats; v.h_stats = c->h_stats;
	v.stream = c->stream; v.copy_stream = c->copy_stream;
	v.d_idx = &c->d_idx; v.idx_slots = &c->idx_slots; v.idx_n = &c->idx_n;
	v.gx = &c->gx;
	return v;
}

int sdti::sync_stats(sdt_ctx *c) { return ::sync_stats(c); }

// the first-occurrence ordinals have done their work (the device has the visiting order): 8 bytes per table slot go back to the arena
int sdti::drop_first(sdt_ctx *c)
{
	HIPCHK(hipSetDevice(c->device));
	if (c->d_first) { HIPCHK(hipFree(c->d_first)); c->d_first = nullptr; }
	return SDT_OK;
}

int sdti::release_pass1(sdt_ctx *c)
{
	const int rc = ::sync_stats(c);
	if (rc != SDT_OK) return rc;
	if (!sdt_tuning_env("SDT_KEEP_POOLS")) {                         // (measurement switch)
		sk_free(c);
	}
	return SDT_OK;
}
// pageable host memory <-> device in pieces through pinned staging buffers: a few threads copy between the caller's
// memory and the staging buffers while the copy engine moves the neighbouring pieces (a plain hipMemcpy of pageable memory
// runs at a third of the link)
// the staging buffers live as long as the process (pinning 64 MiB costs milliseconds: a large export makes dozens of transfers)
struct BigStage {
	static constexpr int NB = 4;
	static constexpr size_t CH = (size_t)32 << 20;
	void *pin[NB] = {};
	hipEvent_t done[NB] = {};
	bool ok = false;
	std::mutex mu;
	bool init()
	{
		if (ok) return true;
		for (int i = 0; i < NB; i++)
			if (hipHostMalloc(&pin[i], CH, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
				return false;
		ok = true;
		return true;
	}
};
static BigStage g_stage;

static void host_copy_mt(void *d, const void *s, size_t n)
{
	constexpr int T = 8;
	if (n < ((size_t)4 << 20)) { memcpy(d, s, n); return; }
	std::thread th[T];
	for (int t = 0; t < T; t++) {
		const size_t a0 = n * t / T, a1 = n * (t + 1) / T;
		th[t] = std::thread([=] { memcpy((char *)d + a0, (const char *)s + a0, a1 - a0); });
	}
	for (int t = 0; t < T; t++) th[t].join();
}

static int big_copy(hipStream_t copy_stream, void *dst, const void *src, size_t bytes, bool to_device)
{
	constexpr int NB = BigStage::NB;
	const size_t CH = BigStage::CH;
	if (bytes < CH) {
		if (hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
		    hipStreamSynchronize(copy_stream) != hipSuccess)
			return fail(SDT_EHIP, "copy of %zu bytes failed", bytes);
		return SDT_OK;
	}
	std::lock_guard<std::mutex> lock(g_stage.mu);
	if (!g_stage.init()) return fail(SDT_ENOMEM, "pinned staging for a %zu-byte transfer", bytes);
	void **pin = g_stage.pin;
	hipEvent_t *done = g_stage.done;
	int rc = SDT_OK;
	const size_t npieces = (bytes + CH - 1) / CH;
	if (to_device) {
		bool used[NB] = {};
		for (size_t k = 0; k < npieces && rc == SDT_OK; k++) {
			const int b = (int)(k % NB);
			const size_t off = k * CH, n = bytes - off < CH ? bytes - off : CH;
			if (used[b] && hipEventSynchronize(done[b]) != hipSuccess) { rc = fail(SDT_EHIP, "upload: event wait failed"); break; }
			host_copy_mt(pin[b], (const char *)src + off, n);
			if (hipMemcpyAsync((char *)dst + off, pin[b], n, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
			    hipEventRecord(done[b], copy_stream) != hipSuccess) { rc = fail(SDT_EHIP, "upload: copy failed"); break; }
			used[b] = true;
		}
	} else {
		// pieces k .. k + NB - 2 are on the link while the threads drain piece k - 1
		for (size_t k = 0; k < npieces + NB - 1 && rc == SDT_OK; k++) {
			if (k >= (size_t)(NB - 1)) {
				const size_t j = k - (NB - 1);
				const int b = (int)(j % NB);
				const size_t off = j * CH, n = bytes - off < CH ? bytes - off : CH;
				if (hipEventSynchronize(done[b]) != hipSuccess) { rc = fail(SDT_EHIP, "download: event wait failed"); break; }
				host_copy_mt((char *)dst + off, pin[b], n);
			}
			if (k < npieces) {
				const int b = (int)(k % NB);
				const size_t off = k * CH, n = bytes - off < CH ? bytes - off : CH;
				if (hipMemcpyAsync(pin[b], (const char *)src + off, n, hipMemcpyDeviceToHost, copy_stream) != hipSuccess ||
				    hipEventRecord(done[b], copy_stream) != hipSuccess) { rc = fail(SDT_EHIP, "download: copy failed"); break; }
			}
		}
	}
	if (hipStreamSynchronize(copy_stream) != hipSuccess && rc == SDT_OK) rc = fail(SDT_EHIP, "transfer: sync failed");
	return rc;
}

int sdti::h2d_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes) { return big_copy(copy_stream, dst, src, bytes, true); }
int sdti::d2h_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes) { return big_copy(copy_stream, dst, src, bytes, false); }
