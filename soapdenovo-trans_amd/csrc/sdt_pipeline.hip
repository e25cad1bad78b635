// sdt_pipeline.hip -- host side of the locality pipeline (design: sdt_superkmer.cuh): chunk pools, the level-1 scatter of reads into
// minimizer buckets, the level-2 split, the count stage's work plan (sdt_count_plan.h) and its launches.  Replaces the reference's
// chop-then-insert worker loop (prlHashReads.c:65-124, 523-526, 600-606).
#include "sdt_ctx.hpp"
#include "sdt_tile.cuh"
#include "sdt_superkmer_kernels.cuh"
#include "sdt_count_plan.h"
#include "sdt_pipeline.hpp"

// ------------------------------------------------------------------------------------------------
// locality pipeline (sdt_superkmer.cuh): scatter super-k-mers -> split -> count in LDS -> merge
// ------------------------------------------------------------------------------------------------
void sk_free(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	void *dev[] = {k.p1.recs, k.p1.meta, k.p1.next, k.p2.recs, k.p2.meta, k.p2.next, k.cursors, k.blk, k.cnt1, k.off1, k.fill1, k.list1,
	               k.cnt2, k.off2, k.fill2, k.list2, k.kmers2, k.kpre2, k.items, k.citems, k.next_item};
	for (void *p : dev)
		if (p) (void)hipFree(p);
	void *host[] = {k.h_off1, k.h_off2, k.h_kpre2, k.h_items, k.h_citems};
	for (void *p : host)
		if (p) (void)hipHostFree(p);
	const uint64_t in_total = k.l2_in_total;         // (the conservation totals belong to the run, not to the pools)
	const bool exchanged = k.exchanged;
	const uint32_t stream_flushes = k.stream_flushes;
	k = sdt_ctx::SkState();
	k.l2_in_total = in_total;
	k.exchanged = exchanged;
	k.stream_flushes = stream_flushes;
}

// LDS bytes of the level-1 scatter for a maximum read length
SkGeo sk_geo(int K, uint64_t max_read_len)
{
	SkGeo g;
	g.mtw = (int)(((uint64_t)SK_TILE_READS * max_read_len + 16 + 15) / 16) + TAIL_PAD + 1;
	g.tile_words = (int)((tile_smem_bytes(g.mtw) / sizeof(uint32_t) + 1) & ~(size_t)1);
	g.hv_words = (int)((SK_TILE_READS * max_read_len + 16 + 1) & ~(uint64_t)1);
	const uint64_t nk_max = (uint64_t)SK_TILE_READS * (max_read_len - K + 1);
	g.bits_words = (int)(nk_max / 64 + 2);
	// long windows (K - m + 1 > 49: the strip kernel's sparse table of window minima) ping-pong between two hash arrays
	g.hv2_words = K - sk_minimizer_len(K) + 1 > 49 ? g.hv_words : 0;
	g.smem = (size_t)g.tile_words * 4 + (size_t)SK_NB1 * 8 + (size_t)(g.hv_words + g.hv2_words) * 4 + (size_t)g.bits_words * 8 + (size_t)(g.bits_words + 2) * 4;
	return g;
}

template <int NW, bool TRACK> static size_t sk_count_smem()
{
	using G = SkCntGeo<NW, TRACK>;
	constexpr int SLOTS = G::SLOTS, BW = SkFmt<NW>::BW, TR = G::TILE;
	// keys (+ ordinals), headers, 5 field words per slot, weights, the prefix / map / index region (= dedupe table), the tile's bases
	return (size_t)(NW + (TRACK ? 1 : 0)) * SLOTS * 8 + (size_t)TR * 8 + (size_t)SLOTS * 20 + (size_t)TR * 4 + G::REGION +
	       (size_t)(LDS_LEAD + TR * BW * 2 + TAIL_PAD) * 4;
}

bool sk_applicable(const sdt_ctx *c, uint64_t max_read_len)
{
	if (c->flags & SDT_FLAG_CONTIG_INDEX)
		return false;
	if (max_read_len < (uint64_t)c->K + 1 || max_read_len > (uint64_t)SK_MAX_READ_LEN)
		return false;
	return sk_geo(c->K, max_read_len).smem <= 160 * 1024;
}

// pool 1 empty, every workgroup without an open chunk
int sk_reset_pool1(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	HIPCHK(hipMemsetAsync(k.p1.next, 0, 4, c->stream));
	HIPCHK(hipMemsetAsync(k.cnt1, 0, SK_NB1 * 4, c->stream));
	hipLaunchKernelGGL(k_sk_init_cursors, dim3(256), dim3(256), 0, c->stream, k.cursors, k.wgs * (uint32_t)SK_NB1, (uint32_t)SK_CAP1, k.blk, k.wgs);
	HIPCHK(hipGetLastError());
	k.pending_kmers = 0;
	return SDT_OK;
}

int sk_alloc(sdt_ctx *c, uint64_t want_kmers, uint64_t per_read)
{
	sdt_ctx::SkState &k = c->sk;
	if (want_kmers > SK_BATCH_MAX_KMERS)
		want_kmers = SK_BATCH_MAX_KMERS;
	if (k.ready && (k.cap_kmers >= want_kmers || k.cap_is_max || k.pending_kmers))
		return SDT_OK;                               // (pools that hold records are never replaced: they are flushed first)
	if (k.ready) {
		HIPCHK(hipStreamSynchronize(c->stream));
		sk_free(c);
	}
	const int rw = sk_rec_words(c->nw), rw2 = sk_rec2_stride(c->nw);      // words per record; per slot of a level-2 chunk
	const int w = c->K - sk_minimizer_len(c->K) + 1;
	// records: a run ends where the minimizer's bucket changes (every (w + 1) / 2 k-mers for a random order of the m-mers) or
	// where the record is full (every `max run` k-mers at the latest): 1 / (2 / (w + 1) + 1 / max run) k-mers per record is what
	// the pools are sized for.  Measured: 10.2 k-mers per record against 8.2 from this formula at K = 31, 23.9 against 19 at
	// K = 63, 6.5 against 5.7 at K = 23 -- the margin IS the head room (a record that finds no chunk takes the direct path,
	// `pool_direct` in the pipeline statistics: nothing is lost, but a fifth of the k-mers of a K = 95 job went that way and
	// tripled its scatter time when the pools were sized at (w + 1) / 2 * 3 / 4).  4-word keys and reads of more than 256 k-mers go
	// through the strip kernel, which also cuts at multiples of the record capacity: a fifth more room.  A read is at least one record.
	const double rate = 2.0 / (double)(w + 1) + 1.0 / (double)sk_max_run(c->K, c->nw);
	double run = 1.0 / rate * ((c->nw == 4 || per_read > (uint64_t)SK_SEQ_MAX_KMERS) ? 0.8 : 1.0);
	if (run > (double)per_read) run = (double)per_read;
	uint64_t div = (uint64_t)run;
	if (div < 2) div = 2;
	div = (uint64_t)clamp_int(sdt_knob_int(sdt_tuning_env("SDT_SK_POOL_DIV"), (int)div), 1, 1 << 20);
	const int mem_pct = sdt_knob_int(sdt_tuning_env("SDT_SK_POOL_MEM_PCT"), 70);
	size_t free_b = 0, total_b = 0;
	HIPCHK(sdti::mem_info(&free_b, &total_b));
	uint64_t cap = want_kmers < (1ULL << 24) ? (1ULL << 24) : want_kmers;
	k.cap_is_max = cap >= SK_BATCH_MAX_KMERS;
	const uint32_t wgs = (uint32_t)c->cu_count * 6;
	for (;; cap /= 2) {
		const uint64_t recs = cap / div;
		// (SDT_SK_POOL_CHUNKS1: test hook -- a level-1 pool of that many chunks, so that small inputs overflow it: tests/test_gpu_parity.py,
		// tests/test_sharded.py)
		const uint64_t chunks1 = sdt_knob_int(sdt_test_env("SDT_SK_POOL_CHUNKS1"), 0) > 0 ? (uint64_t)sdt_knob_int(sdt_test_env("SDT_SK_POOL_CHUNKS1"), 0) : recs / SK_CAP1 + (uint64_t)wgs * SK_NB1 + 1024;
		const uint64_t items = chunks1 / SK_ITEM_CHUNKS + SK_NB1 + 1;
		const uint64_t chunks2 = chunks1 * (SK_CAP1 / SK_CAP2) + items * (SK_NB2 + SK_BLK2) + 1024;      // (+ what an item leaves of its last block of ids)
		const uint64_t bytes = chunks1 * SK_CAP1 * rw * 8 + chunks2 * SK_CAP2 * rw2 * 8 + (chunks1 + chunks2) * 8;
		// (a sharded context adds two send and two receive buffers of pool-1 size: shard_alloc)
		const uint64_t all = c->comm.nranks > 1 ? bytes + chunks1 * SK_CAP1 * rw * 8 * 9 / 2 : bytes;
		if (chunks2 >= (1ULL << SK_LIST2_FILL_SHIFT) - 1 || all > free_b / 100 * (uint64_t)mem_pct) {      // (a list2 entry has 28 bits for the chunk id; all ones = no chunk)
			k.cap_is_max = true;
			if (cap <= (1ULL << 24))
				return fail(SDT_ENOMEM, "super-k-mer pools: %llu MiB needed for the smallest batch, %zu MiB free",
				            (unsigned long long)(bytes >> 20), free_b >> 20);
			continue;
		}
		k.p1.chunks = (uint32_t)chunks1;
		k.p2.chunks = (uint32_t)chunks2;
		k.items_cap = (uint32_t)items;
		break;
	}
	k.wgs = wgs;
	const double t_alloc0 = comm_now();
	HIPCHK(hipMalloc((void **)&k.p1.recs, (size_t)k.p1.chunks * SK_CAP1 * rw * 8));
	HIPCHK(hipMalloc((void **)&k.p1.meta, (size_t)k.p1.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.p1.next, 64));
	HIPCHK(hipMalloc((void **)&k.p2.recs, (size_t)k.p2.chunks * SK_CAP2 * rw2 * 8));
	HIPCHK(hipMalloc((void **)&k.p2.meta, (size_t)k.p2.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.p2.next, 64));
	HIPCHK(hipMalloc((void **)&k.cursors, (size_t)wgs * SK_NB1 * 8));
	HIPCHK(hipMalloc((void **)&k.blk, (size_t)wgs * 8));
	HIPCHK(hipMalloc((void **)&k.cnt1, SK_NB1 * 4));
	HIPCHK(hipMalloc((void **)&k.off1, (SK_NB1 + 1) * 4));
	HIPCHK(hipMalloc((void **)&k.fill1, SK_NB1 * 4));
	HIPCHK(hipMalloc((void **)&k.list1, (size_t)k.p1.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.cnt2, SK_NBF * 4));
	HIPCHK(hipMalloc((void **)&k.off2, (SK_NBF + 1) * 4));
	HIPCHK(hipMalloc((void **)&k.fill2, SK_NBF * 4));
	HIPCHK(hipMalloc((void **)&k.list2, (size_t)k.p2.chunks * 4));
	HIPCHK(hipMalloc((void **)&k.kmers2, SK_NBF * 8));
	HIPCHK(hipMalloc((void **)&k.kpre2, (SK_NBF + 1) * 8));
	HIPCHK(hipMalloc((void **)&k.items, (size_t)k.items_cap * sizeof(SkItem)));
	HIPCHK(hipHostMalloc((void **)&k.h_off1, (SK_NB1 + 1) * 4, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_off2, (SK_NBF + 1) * 4, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_kpre2, (SK_NBF + 1) * 8, hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void **)&k.h_items, (size_t)k.items_cap * sizeof(SkItem), hipHostMallocDefault));
	k.citems_cap = (uint32_t)SK_NBF + k.p2.chunks / SK_COUNT_ITEM_CHUNKS + 1;
	HIPCHK(hipMalloc((void **)&k.citems, (size_t)k.citems_cap * sizeof(uint4)));
	HIPCHK(hipHostMalloc((void **)&k.h_citems, (size_t)k.citems_cap * sizeof(uint4), hipHostMallocDefault));
	HIPCHK(hipMalloc((void **)&k.next_item, SK_MAX_COUNT_LAUNCHES * sizeof(uint32_t)));
	k.cap_kmers = cap;
	k.ready = true;
	if (sdt_env("SDT_TIMING"))
		fprintf(stderr, "[libsdt_gpu] super-k-mer pools for %llu k-mers per batch: %.1f GiB in %.0f ms\n", (unsigned long long)cap,
		        ((double)k.p1.chunks * SK_CAP1 * rw + (double)k.p2.chunks * SK_CAP2 * rw2) * 8 / (1 << 30), (comm_now() - t_alloc0) * 1e3);
	return sk_reset_pool1(c);
}

template <int NW, bool TRACK> static int sk_launch_count_t(sdt_ctx *c, uint32_t i0, uint32_t i1, uint32_t launch)
{
	sdt_ctx::SkState &k = c->sk;
	const size_t smem = sk_count_smem<NW, TRACK>();
	// persistent workgroups: as many as the LDS tables let the chip hold; they take work items first come first served
	const unsigned per_cu = (unsigned)((160 * 1024) / (smem + 256));
	unsigned grid = (unsigned)c->cu_count * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
	if (grid > i1 - i0) grid = i1 - i0;
	HIPCHK(hipFuncSetAttribute((const void *)k_sk_count<NW, TRACK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
	hipLaunchKernelGGL((k_sk_count<NW, TRACK>), dim3(grid), dim3(SkCntGeo<NW, TRACK>::TPB), smem, c->stream, k.p2, k.list2, (const uint4 *)k.citems, i0, i1,
	                   k.next_item + launch, c->K, flat_of<NW>(c), c->d_stats);
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

template <int NW> static int sk_launch_count(sdt_ctx *c, uint32_t i0, uint32_t i1, uint32_t launch)
{
	return c->d_first ? sk_launch_count_t<NW, true>(c, i0, i1, launch) : sk_launch_count_t<NW, false>(c, i0, i1, launch);
}


// level 1 done: close the open chunks, list the chunk ids bucket by bucket; h_off1 is valid on return (host sync)
int sk_list1(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	const int g = c->cu_count * 8;
	hipLaunchKernelGGL(k_sk_seal, dim3(256), dim3(256), 0, c->stream, k.cursors, k.wgs * (uint32_t)SK_NB1, k.blk, k.wgs, k.p1, (uint32_t)SK_CAP1);
	hipLaunchKernelGGL(k_sk_scan, dim3(1), dim3(1024), 0, c->stream, k.cnt1, k.off1, k.fill1, (int)SK_NB1, (const unsigned long long *)nullptr, (unsigned long long *)nullptr);
	hipLaunchKernelGGL(k_sk_chunk_place_few, dim3(g), dim3(256), 0, c->stream, k.p1, k.off1, k.fill1, k.list1, (int)SK_NB1);
	SK_CHK(hipGetLastError());
	SK_CHK(hipMemcpyAsync(k.h_off1, k.off1, (SK_NB1 + 1) * 4, hipMemcpyDeviceToHost, c->stream));
	{ const int rcw = c->comm.sync_watched(c->stream, "the chunk lists of a round"); if (rcw != SDT_OK) return rcw; }
	k.st_chunks1 = k.h_off1[SK_NB1];
	return SDT_OK;
}

// level 2: the nitems work items in k.h_items (runs of chunks of `src` named by `list`) are split into pool 2, whose chunks
// are then listed per final bucket; asynchronous (the lists are read back by sk_count_all).  `after_l2`: recorded once
// the records have left `src`.
int sk_split(sdt_ctx *c, const SkPool &src, const uint32_t *list, uint32_t nitems, hipEvent_t after_l2)
{
	sdt_ctx::SkState &k = c->sk;
	const int g = c->cu_count * 8;
	SK_CHK(hipMemsetAsync(k.p2.next, 0, 4, c->stream));
	SK_CHK(hipMemsetAsync(k.kmers2, 0, SK_NBF * 8, c->stream));
	SK_CHK(hipMemsetAsync(k.cnt2, 0, SK_NBF * 4, c->stream));
	if (nitems) {
		SK_CHK(hipMemcpyAsync(k.items, k.h_items, (size_t)nitems * sizeof(SkItem), hipMemcpyHostToDevice, c->stream));
		// staged (round 5): records wait in LDS for a group of 4 (2), groups are stored whole; one workgroup of 1024 lanes per CU by its LDS alone
		const size_t sm = c->nw == 1 ? SkL2Stage<1>::SMEM : (c->nw == 2 ? SkL2Stage<2>::SMEM : SkL2Stage<4>::SMEM);
		const void *fn = c->nw == 1 ? (const void *)k_sk_scatter_records_staged<1> : (c->nw == 2 ? (const void *)k_sk_scatter_records_staged<2> : (const void *)k_sk_scatter_records_staged<4>);
		SK_CHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
		if (c->nw == 1) hipLaunchKernelGGL(k_sk_scatter_records_staged<1>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		else if (c->nw == 2) hipLaunchKernelGGL(k_sk_scatter_records_staged<2>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		else hipLaunchKernelGGL(k_sk_scatter_records_staged<4>, dim3(nitems), dim3(SK_L2S_TPB), sm, c->stream, src, list, k.items, k.p2, k.cnt2, k.kmers2, c->d_stats);
		SK_CHK(hipGetLastError());
	}
	if (after_l2)
		SK_CHK(hipEventRecord(after_l2, c->stream));
	hipLaunchKernelGGL(k_sk_scan, dim3(1), dim3(1024), 0, c->stream, k.cnt2, k.off2, k.fill2, (int)SK_NBF, (const unsigned long long *)k.kmers2, k.kpre2);
	hipLaunchKernelGGL(k_sk_chunk_place, dim3(g), dim3(256), 0, c->stream, k.p2, k.off2, k.fill2, k.list2);
	SK_CHK(hipGetLastError());
	SK_CHK(hipMemcpyAsync(k.h_kpre2, k.kpre2, (SK_NBF + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	SK_CHK(hipMemcpyAsync(k.h_off2, k.off2, (SK_NBF + 1) * 4, hipMemcpyDeviceToHost, c->stream));
	return SDT_OK;
}

// count pool 2 bucket by bucket (host sync first: the chunk lists of sk_split come back)
int sk_count_all(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	SK_CHK(hipStreamSynchronize(c->stream));
	k.st_chunks2 = k.h_off2[SK_NBF];
	k.st_flushes++;
	k.stream_flushes++;
	k.l2_in_total += k.h_kpre2[SK_NBF];
	// work items = pieces of buckets of at most SK_COUNT_ITEM_CHUNKS chunks; launches of at most SK_COUNT_KMERS
	// k-mers (every one might be a new node: ensure_room)
	int rc = SDT_OK;
	uint32_t nci = 0;
	SK_CHK(hipMemsetAsync(k.next_item, 0, SK_MAX_COUNT_LAUNCHES * sizeof(uint32_t), c->stream));
	// A launch must find room for every node it may create.  "Every occurrence is a new node" is hopeless for a batch of
	// 2^33 k-mers, so: a first launch of at most 2^26 k-mers under that bound, then launches bounded by twice the rate of
	// new nodes per occurrence seen so far (later data usually brings fewer new nodes, not more; should it bring more, the load
	// factor suffers until the next look but the table cannot fill: see the 95 % rule below).
	// (the items and launches are a pure function of the chunk lists: sdt_count_plan.h, tested on the CPU)
	std::vector<uint32_t> first_item(SK_MAX_COUNT_LAUNCHES + 2);   // first item of every launch
	std::vector<uint64_t> launch_kmers(SK_MAX_COUNT_LAUNCHES + 2);
	uint32_t nlaunches = 0;
	
	if (!sk_plan_count_items(k.h_off2, (const uint64_t *)k.h_kpre2, (uint32_t)SK_NBF, c->kmers_known == 0 ? (1ULL << 26) : SK_COUNT_KMERS, SK_COUNT_KMERS,
	                         SK_MAX_COUNT_LAUNCHES, SK_COUNT_PACK_CHUNKS, SK_COUNT_ITEM_CHUNKS, (uint32_t *)k.h_citems, k.citems_cap,
	                         first_item.data(), launch_kmers.data(), (uint32_t)first_item.size(), &nci, &nlaunches))
		return fail(SDT_ESTATE, "count stage: work item table overflow");
	first_item.resize(nlaunches + 1);
	launch_kmers.resize(nlaunches);
	std::vector<uint32_t> sort_tmp;
	auto guess_of = [&](uint64_t kmers) -> uint64_t {
		uint64_t bound = kmers;
		if (c->kmers_known) {
			const double rate = (double)c->distinct_known / (double)c->kmers_known;
			const uint64_t guess = (uint64_t)((double)kmers * (2.0 * rate < 1.0 ? 2.0 * rate : 1.0)) + (1ULL << 22);
			if (guess < bound) bound = guess;
		}
		return bound;
	};
	const size_t nl = first_item.size() - 1;
	for (size_t l = 0; l < nl && rc == SDT_OK;) {
		const uint32_t i0 = first_item[l];
		if (i0 == first_item[l + 1]) {
			l++;
			continue;
		}
		if (c->kmers_known == 0 && l > 0) {
			rc = sync_stats(c);                      // the first launch has run: its rate of new nodes bounds the rest
			if (rc != SDT_OK) break;
		}
		uint64_t bound = guess_of(launch_kmers[l]), hard = launch_kmers[l];
		rc = ensure_room(c, bound);
		// the guess keeps the load factor; this keeps the table from FILLING should the guess be wrong: whatever the data,
		// the nodes known + every k-mer launched since + this launch must fit 95 % of the slots
		if (rc == SDT_OK && (double)(c->distinct_known + c->hard_since_sync + launch_kmers[l]) > 0.95 * (double)c->slots) {
			rc = sync_stats(c);
			if (rc == SDT_OK && (double)(c->distinct_known + launch_kmers[l]) > 0.95 * (double)c->slots)
				rc = grow_table(c, c->distinct_known + launch_kmers[l]);
		}
		// The planned launches behind this one join it as long as neither rule would have to look at the device's counters for
		// them: a launch boundary is a drained GPU (every workgroup waits for the slowest), and it is only needed where the host
		// decides about the table.  (45 planned launches per step of the 200 M-read workload become about a dozen.)
		size_t m = l;
		while (rc == SDT_OK && c->kmers_known && m + 1 < nl) {
			const uint64_t b2 = guess_of(launch_kmers[m + 1]);
			if ((double)(c->distinct_known + c->kmers_since_sync + bound + b2) > (double)c->slots * MAX_LOAD)
				break;
			if ((double)(c->distinct_known + c->hard_since_sync + hard + launch_kmers[m + 1]) > 0.95 * (double)c->slots)
				break;
			bound += b2;
			hard += launch_kmers[m + 1];
			m++;
		}
		const uint32_t i1 = first_item[m + 1];
		// (largest first over everything this launch hands out -- the plan did it per planned launch; the sort is stable, so the
		// concatenation of sorted runs comes out as one)
		if (m > l)
			sk_plan_largest_first((uint32_t *)k.h_citems, i0, i1, sort_tmp);
		if (rc == SDT_OK)
			SK_CHK(hipMemcpyAsync(k.citems + i0, k.h_citems + i0, (size_t)(i1 - i0) * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
		if (rc == SDT_OK)
			rc = c->nw == 1 ? sk_launch_count<1>(c, i0, i1, (uint32_t)l) : c->nw == 2 ? sk_launch_count<2>(c, i0, i1, (uint32_t)l) : sk_launch_count<4>(c, i0, i1, (uint32_t)l);
		if (rc == SDT_OK) {                              // (only what was launched counts)
			c->kmers_since_sync += bound;
			c->hard_since_sync += hard;
		}
		l = m + 1;
	}
	// (the pinned item list must outlive its copy: the next flush rewrites it only after this stream has drained)
	return rc;
}

// work items of level 2 over the chunks [lo, hi) of bucket b in a list: at most SK_ITEM_CHUNKS chunks each
int sk_add_items(sdt_ctx::SkState &k, uint32_t &nitems, uint32_t b, uint32_t lo, uint32_t hi)
{
	for (uint32_t c0 = lo; c0 < hi; c0 += SK_ITEM_CHUNKS) {
		if (nitems >= k.items_cap)
			return fail(SDT_EHIP, "super-k-mer pipeline: item table overflow");
		k.h_items[nitems++] = SkItem{b, c0, c0 + SK_ITEM_CHUNKS < hi ? c0 + SK_ITEM_CHUNKS : hi, 0};
	}
	return SDT_OK;
}

int sk_flush_sharded(sdt_ctx *c);

// everything scattered so far goes into the node table: seal + list the level-1 chunks, split every level-1 bucket,
// list the level-2 chunks, count bucket by bucket
int sk_flush(sdt_ctx *c)
{
	sdt_ctx::SkState &k = c->sk;
	if (!k.ready || k.pending_kmers == 0 || k.flushing)
		return SDT_OK;
	if (c->comm.nranks > 1)
		return fail(SDT_ESTATE, "sharded context: the pipeline is drained by the collective calls (sdt_gpu_count_reads_sharded)");
	k.flushing = true;
	EventPair *ev = next_event(c), *ev2 = next_event(c);
	if (!ev || !ev2) { k.flushing = false; return fail(SDT_EHIP, "hipEventCreate failed"); }
	ev = ev2 - 1;                                    // next_event may have moved the vector
	ev->kmers = ev2->kmers = 0;
	ev->stage = SDT_STAGE_SK_SPLIT;
	ev2->stage = SDT_STAGE_SK_COUNT;
	int rc = hipEventRecord(ev->a, c->stream) == hipSuccess ? SDT_OK : fail(SDT_EHIP, "hipEventRecord failed");
	if (rc == SDT_OK) rc = sk_list1(c);
	uint32_t nitems = 0;
	for (uint32_t b = 0; b < (uint32_t)SK_NB1 && rc == SDT_OK; b++)
		rc = sk_add_items(k, nitems, b, k.h_off1[b], k.h_off1[b + 1]);
	if (rc == SDT_OK) rc = sk_split(c, k.p1, k.list1, nitems, nullptr);
	// pool 1 is free again: the next batch may scatter while this one is counted (same stream: in order)
	if (rc == SDT_OK) rc = sk_reset_pool1(c);
	if (rc == SDT_OK && (hipEventRecord(ev->b, c->stream) != hipSuccess || hipEventRecord(ev2->a, c->stream) != hipSuccess))
		rc = fail(SDT_EHIP, "hipEventRecord failed");
	if (rc == SDT_OK) {
		k.pending_kmers = 0;
		rc = sk_count_all(c);
	}
	if (hipEventRecord(ev2->b, c->stream) != hipSuccess && rc == SDT_OK)
		rc = fail(SDT_EHIP, "hipEventRecord failed");
	k.flushing = false;
	return rc;
}

template <int NW> static Table<NW> sk_tbl(const sdt_ctx *c, bool allow_direct)
{
	Table<NW> t = flat_of<NW>(c);
	if (!allow_direct)
		t.ent = nullptr;
	return t;
}

// one launch of the level-1 scatter over reads [0, nr) of a device-resident batch (ordinals from `ob`)
// (allow_direct = false: a record that finds no chunk is an error, not a put into the local table -- sharded contexts, where
// the local table owns only some buckets, and the range-weighing sample, whose reads are scattered a second time)
int sk_scatter_launch(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nr, uint64_t max_read_len, uint64_t ob,
                             bool allow_direct)
{
	sdt_ctx::SkState &k = c->sk;
	const uint64_t per_read = max_read_len - c->K + 1;
	const SkGeo geo = sk_geo(c->K, max_read_len);
	const int m = sk_minimizer_len(c->K), ncap = sk_max_run(c->K, c->nw);
	const uint64_t ntiles = (nr + SK_TILE_READS - 1) / SK_TILE_READS;
	const unsigned grid = (unsigned)(ntiles < k.wgs ? ntiles : k.wgs);
	EventPair *ev = next_event(c);
	if (!ev)
		return fail(SDT_EHIP, "hipEventCreate failed");
	ev->kmers = nr * per_read;
	ev->stage = SDT_STAGE_SK_SCATTER;
	HIPCHK(hipEventRecord(ev->a, c->stream));
#define SK_SCATTER(NW)                                                                                                             \
	do {                                                                                                                       \
		HIPCHK(hipFuncSetAttribute((const void *)k_sk_scatter_reads<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.smem)); \
		hipLaunchKernelGGL(k_sk_scatter_reads<NW>, dim3(grid), dim3(TPB), geo.smem, c->stream, d_words, d_offs, nr, c->K, m, ncap, \
		                   geo.mtw, geo.tile_words, geo.hv_words, geo.hv2_words, geo.bits_words, k.p1, k.cursors, k.blk, k.cnt1, sk_tbl<NW>(c, allow_direct), c->d_stats, ob, c->ord_stride); \
	} while (0)
	// one lane per read where the window length has an instantiation and the run list can hold a read
	const int w = c->K - m + 1;
	static const bool no_seq = sdt_tuning_env("SDT_SK_STRIPS") != NULL;          // A/B switch (tools/)
	// (instantiated: every odd window of 1-word keys with K >= 17 and of 2-word keys, i.e. every odd K from 17 to 63)
	const bool seq1 = c->nw == 1 && (w & 1) && w >= 9 && w <= 21 && per_read <= (uint64_t)SK_SEQ_MAX_KMERS;
	const bool seq2 = c->nw == 2 && (w & 1) && w >= 23 && w <= 53 && per_read <= (uint64_t)SK_SEQ_MAX_KMERS;
	if ((seq1 || seq2) && !no_seq) {
		SkSeqLaunch a;
		a.words = d_words; a.offs = d_offs; a.nreads = nr; a.K = c->K; a.m = m; a.ncap = ncap;
		a.mtw = (int)(((uint64_t)SK_SEQ_TILE * max_read_len + 16 + 15) / 16) + TAIL_PAD + 1;
		a.pool = k.p1; a.cursors = k.cursors; a.blk = k.blk; a.cnt = k.cnt1; a.stats = c->d_stats;
		a.ord_base = ob; a.ord_stride = c->ord_stride; a.max_wgs = k.wgs; a.cu_count = c->cu_count; a.stream = c->stream;
		HIPCHK(seq1 ? sk_seq_launch_nw1(w, a, sk_tbl<1>(c, allow_direct)) : (w <= 33 ? sk_seq_launch_nw2_lo(w, a, sk_tbl<2>(c, allow_direct))
		            : (w <= 43 ? sk_seq_launch_nw2_mid(w, a, sk_tbl<2>(c, allow_direct)) : sk_seq_launch_nw2_hi(w, a, sk_tbl<2>(c, allow_direct)))));
	} else if (c->nw == 1) SK_SCATTER(1);
	else if (c->nw == 2) SK_SCATTER(2);
	else SK_SCATTER(4);
#undef SK_SCATTER
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(ev->b, c->stream));
	k.pending_kmers += nr * per_read;
	return SDT_OK;
}

// k-mers a batch may hold before it is flushed: the pools' capacity -- except for the first batches of a stream whose length
// the caller has announced (sdt_gpu_hint_total_kmers): 1/16 of the job, then 1/8, 1/4 ... .  The kernels need about twice the
// time of the copies, so a batch's copies hide behind the counting of the batches before it as long as it is at most about
// twice their size; one large batch after a small first one left the GPU waiting for 9 GB of copies (measured: 66 instead
// of 68 G k-mers/s from host memory), equal quarters of the job merge more often than they must.
uint64_t sk_batch_limit(const sdt_ctx *c)
{
	const sdt_ctx::SkState &k = c->sk;
	if (c->expect_kmers && c->expect_kmers / 16 >= (1ULL << 27) && k.stream_flushes < 4) {
		const uint64_t lim = (c->expect_kmers / 16) << k.stream_flushes;
		if (lim < k.cap_kmers)
			return lim;
	}
	return k.cap_kmers;
}

// chop + scatter a device-resident batch into the level-1 buckets (flushing whenever the pools are full)
int sk_scatter(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads, uint64_t max_read_len)
{
	sdt_ctx::SkState &k = c->sk;
	const uint64_t per_read = max_read_len - c->K + 1;
	int rc = SDT_OK;
	if (k.ready && k.pending_kmers && !k.cap_is_max && k.cap_kmers < k.pending_kmers + nreads * per_read && k.cap_kmers < (1ULL << 31))
		rc = sk_flush(c);                            // the pools are about to be replaced by larger ones
	if (rc == SDT_OK) {
		// without SDT_FLAG_PARTITION the pipeline only runs for jobs past 2^27 k-mers: start with pools for 2^31 at once
		// (sk_alloc halves that until it fits the free memory) instead of growing there batch by batch
		uint64_t want = k.pending_kmers + nreads * per_read;
		if (!(c->flags & SDT_FLAG_PARTITION) && want < (1ULL << 31))
			want = 1ULL << 31;
		// a caller that streams its reads in and has said how much is coming (sdt_gpu_hint_total_kmers): pools for 2^32 k-mers from
		// the first batch on (26 GiB at K = 31), not grown there batch by batch.  NOT pools for the whole job any more: fewer, larger
		// batches merge a little less (quarters of a 14 G k-mer job measured 3 % slower than one batch), but the 103 GiB of pools of
		// that job cost 1.4 - 4.7 s to allocate whenever the box's memory had been used before (sdt_mem.hip) -- a hundred times the gain.
		if (c->expect_kmers > want) {
			const uint64_t lim = 1ULL << clamp_int(sdt_knob_int(sdt_tuning_env("SDT_SK_HINT_BATCH_LOG2"), 32), 24, 34);
			const uint64_t hinted = c->expect_kmers < lim ? c->expect_kmers : lim;
			if (hinted > want) want = hinted;
		}
		rc = sk_alloc(c, want, per_read);
	}
	if (rc != SDT_OK)
		return rc;
	for (uint64_t r0 = 0; r0 < nreads;) {
		if (k.pending_kmers + per_read * SK_TILE_READS > sk_batch_limit(c)) {
			rc = sk_flush(c);
			// a stream that keeps filling SMALL pools gets larger ones: fewer batches = fewer merges per distinct key.  Past 2^31
			// k-mers they stay: replacing tens of GiB was seen to stall for seconds in hipFree / hipMalloc now and then.
			if (rc == SDT_OK && !k.cap_is_max && k.cap_kmers < (1ULL << 31))
				rc = sk_alloc(c, k.cap_kmers * 2, per_read);
			if (rc != SDT_OK)
				return rc;
		}
		uint64_t nr = (sk_batch_limit(c) - k.pending_kmers) / per_read / SK_TILE_READS * SK_TILE_READS;
		if (nr > nreads - r0) nr = nreads - r0;
		rc = sk_scatter_launch(c, d_words, d_offs + r0, nr, max_read_len, c->ord_base + r0 * c->ord_stride);
		if (rc != SDT_OK)
			return rc;
		r0 += nr;
	}
	return SDT_OK;
}

extern "C" {
int sdt_sk_plan_count_items(const uint32_t *off2, const uint64_t *kpre2, uint32_t nbuckets, uint64_t first_limit, uint64_t limit,
                            uint32_t max_launches, uint32_t *items, uint32_t items_cap, uint32_t *first_item, uint64_t *launch_kmers,
                            uint32_t launches_cap, uint32_t *nitems, uint32_t *nlaunches)
{
	if (!off2 || !kpre2 || !items || !first_item || !launch_kmers || !nitems || !nlaunches)
		return fail(SDT_EINVAL, "NULL argument");
	if (max_launches < 1 || limit == 0 || first_limit == 0)
		return fail(SDT_EINVAL, "bad argument (at least one launch, limits of at least one k-mer)");
	if (!sk_plan_count_items(off2, kpre2, nbuckets, first_limit, limit, max_launches, SK_COUNT_PACK_CHUNKS, SK_COUNT_ITEM_CHUNKS, items, items_cap,
	                         first_item, launch_kmers, launches_cap, nitems, nlaunches))
		return fail(SDT_ENOMEM, "output arrays too small");
	return SDT_OK;
}

int sdt_kmer_bucket(const uint64_t *key_words_msw_first, int K)
{
	// the level-1 minimizer bucket (0..255) of a canonical k-mer, as on the device
	if (!key_words_msw_first || K < 13 || K > 127)
		return -1;
	const int nw = K <= 31 ? 1 : (K <= 63 ? 2 : 4), m = sk_minimizer_len(K);
	uint32_t best = 0xFFFFFFFFu;
	for (int p = 0; p + m <= K; p++) {
		uint32_t fw = 0;
		for (int i = 0; i < m; i++) {
			const int bit = 2 * (K - 1 - (p + i));       // base p + i of the k-mer, counted from its low end
			const uint64_t w = key_words_msw_first[nw - 1 - bit / 64];
			fw = (fw << 2) | (uint32_t)((w >> (bit % 64)) & 3u);
		}
		const uint32_t hv = sk_mmer_hash(sk_canon_mmer(fw, m));
		if (hv < best) best = hv;
	}
	return (int)sk_l1_bucket(sk_bucket_hash(best));
}

int sdt_kmer_final_bucket(const uint64_t *key_words_msw_first, int K)
{
	// the final minimizer bucket (0 .. 2^18 - 1) of a k-mer: the unit of the count stage (csrc/sdt_minimizer.cuh,
	// the function the device's look-ups call)
	if (!key_words_msw_first || K < 13 || K > 127)
		return -1;
	if (K <= 31) { Key<1> k{{key_words_msw_first[0]}}; return (int)key_final_bucket<1>(k, K); }
	if (K <= 63) { Key<2> k{{key_words_msw_first[0], key_words_msw_first[1]}}; return (int)key_final_bucket<2>(k, K); }
	Key<4> k{{key_words_msw_first[0], key_words_msw_first[1], key_words_msw_first[2], key_words_msw_first[3]}};
	return (int)key_final_bucket<4>(k, K);
}
#ifdef SDT_SK_L2_LOG
// debug builds only (not declared in include/sdt_gpu.h): point the level-2 scatter's slot log at a device buffer of `cap` words
extern "C" int sdt_gpu_debug_l2_log(sdt_ctx *c, void *d_buf, uint64_t cap)
{
	unsigned long long *p = (unsigned long long *)d_buf, cp = cap;
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_l2_log), &p, sizeof p));
	HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_l2_log_cap), &cp, sizeof cp));
	return SDT_OK;
}
#endif
} // extern "C"
