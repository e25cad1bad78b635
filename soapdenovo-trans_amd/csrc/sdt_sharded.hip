// sdt_sharded.hip -- multi-GPU pass 1: ranks own contiguous ranges of the level-1 minimizer buckets, the level-1 chunks of a round travel
// to their owners (grouped ncclSend / ncclRecv over xGMI, sdt_comm.cuh), every rank splits and counts what it owns.  Reference:
// prlHashReads.c:79-88 routes every k-mer to thread hash_kmer % thrd_num.
#include "sdt_ctx.hpp"
#include "sdt_tile.cuh"
#include "sdt_superkmer_kernels.cuh"
#include "sdt_pipeline.hpp"

// ------------------------------------------------------------------------------------------------
// multi-GPU: ranks own contiguous ranges of the level-1 buckets; level-1 chunks travel to their owner
// ------------------------------------------------------------------------------------------------
void shard_free(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	for (int i = 0; i < 2; i++) {
		if (h.send[i]) (void)hipFree(h.send[i]);
		if (h.recv[i]) (void)hipFree(h.recv[i]);
		if (h.send_meta[i]) (void)hipFree(h.send_meta[i]);
		if (h.recv_meta[i]) (void)hipFree(h.recv_meta[i]);
		if (h.ev_gather[i]) (void)hipEventDestroy(h.ev_gather[i]);
		if (h.ev_xdone[i]) (void)hipEventDestroy(h.ev_xdone[i]);
		if (h.ev_l2[i]) (void)hipEventDestroy(h.ev_l2[i]);
	}
	if (h.iota) (void)hipFree(h.iota);
	h = sdt_ctx::Shard();
}

int shard_alloc(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	if (h.send[0] && h.send_chunks >= k.p1.chunks)
		return SDT_OK;
	HIPCHK(hipStreamSynchronize(c->stream));
	shard_free(c);
	const size_t cw = (size_t)SK_CAP1 * sk_rec_words(c->nw) * 8;
	h.send_chunks = k.p1.chunks;
	h.recv_chunks = k.p1.chunks + k.p1.chunks / 4;   // a rank receives ~ what it sends; head room for unequal buckets
	if (sdt_test_env("SDT_SHARD_RECV_CHUNKS"))             // (tests: force sub-rounds)
		h.recv_chunks = (uint32_t)strtoul(sdt_test_env("SDT_SHARD_RECV_CHUNKS"), nullptr, 10);
	for (int i = 0; i < 2; i++) {
		HIPCHK(hipMalloc((void **)&h.send[i], (size_t)h.send_chunks * cw));
		HIPCHK(hipMalloc((void **)&h.recv[i], (size_t)h.recv_chunks * cw));
		HIPCHK(hipMalloc((void **)&h.send_meta[i], (size_t)h.send_chunks * 4));
		HIPCHK(hipMalloc((void **)&h.recv_meta[i], (size_t)h.recv_chunks * 4));
		HIPCHK(hipEventCreateWithFlags(&h.ev_gather[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h.ev_xdone[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h.ev_l2[i], hipEventDisableTiming));
	}
	HIPCHK(hipMalloc((void **)&h.iota, (size_t)h.recv_chunks * 4));
	hipLaunchKernelGGL(k_sk_iota, dim3(1024), dim3(256), 0, c->stream, h.iota, h.recv_chunks);
	HIPCHK(hipGetLastError());
	return SDT_OK;
}

// split + count what the last exchange delivered
int shard_finish_pending(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	if (!h.pending)
		return SDT_OK;
	h.pending = false;
	const int slot = h.pending_slot;
	EventPair *ev = next_event(c), *ev2 = next_event(c);
	if (!ev || !ev2) return fail(SDT_EHIP, "hipEventCreate failed");
	ev = ev2 - 1;
	ev->kmers = ev2->kmers = 0;
	ev->stage = SDT_STAGE_SK_SPLIT;
	ev2->stage = SDT_STAGE_SK_COUNT;
	HIPCHK(hipStreamWaitEvent(c->stream, h.ev_xdone[slot], 0));
	HIPCHK(hipEventRecord(ev->a, c->stream));
	if (h.items.size() > k.items_cap)
		return fail(SDT_EHIP, "super-k-mer pipeline: item table overflow");
	memcpy(k.h_items, h.items.data(), h.items.size() * sizeof(SkItem));
	SkPool src = {h.recv[slot], h.recv_meta[slot], nullptr, h.recv_chunks};
	int rc = sk_split(c, src, h.iota, (uint32_t)h.items.size(), h.ev_l2[slot]);
	if (rc != SDT_OK) return rc;
	h.l2_recorded[slot] = true;                      // (only an event that was recorded may be waited for)
	HIPCHK(hipEventRecord(ev->b, c->stream));
	HIPCHK(hipEventRecord(ev2->a, c->stream));
	rc = sk_count_all(c);
	HIPCHK(hipEventRecord(ev2->b, c->stream));
	return rc;
}

// COLLECTIVE.  Level-1 chunks scattered since the last call go to the ranks that own their buckets; what the previous
// call's exchange delivered is split and counted meanwhile.  Sub-rounds when a rank would receive more than its buffer holds.
int sk_flush_sharded(sdt_ctx *c)
{
	sdt_ctx::Shard &h = c->sh;
	sdt_ctx::SkState &k = c->sk;
	Comm &cm = c->comm;
	const int n = cm.nranks, me = cm.rank;
	k.exchanged = true;
	int rc = sk_list1(c);
	if (rc != SDT_OK) return rc;
	// everybody's chunk counts per bucket
	std::vector<uint32_t> mat((size_t)n * (SK_NB1 + 1));
	rc = cm.allgather_host(k.h_off1, mat.data(), (SK_NB1 + 1) * sizeof(uint32_t));
	if (rc != SDT_OK) return rc;
	auto M = [&](int r, uint32_t b) { return shard_mat(mat.data(), r, b); };
	const uint32_t *blo = h.ranges;
	// sub-rounds, pieces and buffer layouts: pure functions of the matrix (sdt_shard_plan.h) -- every rank computes every
	// rank's layout from it, so all agree without another message
	const uint32_t S = shard_subrounds(mat.data(), n, blo, h.recv_chunks);
	const size_t cw = (size_t)SK_CAP1 * sk_rec_words(c->nw) * 8;
	for (uint32_t t = 0; t < S; t++) {
		const int slot = (int)(h.round & 1);
		ShardRound sr;
		shard_round(mat.data(), n, me, blo, t, S, sr);
		auto piece = [&](int s2, int d, uint32_t &lo, uint32_t &hi) { shard_piece(mat.data(), blo, s2, d, t, S, lo, hi); };
		SkGatherPlan plan;
		memset(&plan, 0, sizeof plan);
		plan.n = n;
		plan.self = me;
		std::vector<void *> sp(n), rp(n), smp(n), rmp(n);
		std::vector<size_t> sb(n, 0), rb(n, 0), smb(n, 0), rmb(n, 0), oboff((size_t)n * n, 0), obmoff((size_t)n * n, 0);
		const uint32_t send_at = sr.send_total, recv_at = sr.recv_total;
		std::vector<SkItem> cur;                     // level-2 work items of THIS exchange (h.items still describes the last one)
		for (int p = 0; p < n; p++) {
			plan.begin[p] = sr.send_begin[p];
			plan.pre[p + 1] = plan.pre[p] + sr.send_count[p];
			plan.dst0[p] = sr.send_at[p];
			if (p != me) {
				sp[p] = (uint8_t *)h.send[slot] + (size_t)sr.send_at[p] * cw;
				smp[p] = h.send_meta[slot] + sr.send_at[p];
				sb[p] = (size_t)sr.send_count[p] * cw;
				smb[p] = (size_t)sr.send_count[p] * 4;
			}
		}
		for (int s2 = 0; s2 < n; s2++) {             // receive buffer: one run per source, rank order (mine included)
			uint32_t lo, hi;
			piece(s2, me, lo, hi);
			const uint32_t at = sr.recv_at[s2];
			rp[s2] = (uint8_t *)h.recv[slot] + (size_t)at * cw;
			rmp[s2] = h.recv_meta[slot] + at;
			rb[s2] = (size_t)(hi - lo) * cw;
			rmb[s2] = (size_t)(hi - lo) * 4;
			// level-2 work items over this run: its chunks are in bucket order
			for (uint32_t b = blo[me]; b < blo[me + 1] && rc == SDT_OK; b++) {
				const uint32_t x0 = M(s2, b) > lo ? M(s2, b) : lo, x1 = M(s2, b + 1) < hi ? M(s2, b + 1) : hi;
				for (uint32_t c0 = at + (x0 - lo); x1 > x0 && c0 < at + (x1 - lo); c0 += SK_ITEM_CHUNKS) {
					const uint32_t c1 = c0 + SK_ITEM_CHUNKS < at + (x1 - lo) ? c0 + SK_ITEM_CHUNKS : at + (x1 - lo);
					cur.push_back(SkItem{b, c0, c1, 0});
				}
			}
		}
		if (send_at > h.send_chunks || recv_at > h.recv_chunks)
			return fail(SDT_EFULL, "exchange buffers too small: %u / %u chunks to send, %u / %u to receive", send_at, h.send_chunks, recv_at, h.recv_chunks);
		// outbox layout of every rank (shared-memory transport): destinations in rank order
		if (cm.kind == 2)
			for (int s2 = 0; s2 < n; s2++) {
				size_t at = 0, mat_at = 0;
				for (int d = 0; d < n; d++) {
					if (d == s2) continue;
					uint32_t lo, hi;
					piece(s2, d, lo, hi);
					oboff[(size_t)s2 * n + d] = at;
					at += (size_t)(hi - lo) * cw;
				}
				for (int d = 0; d < n; d++) {
					if (d == s2) continue;
					uint32_t lo, hi;
					piece(s2, d, lo, hi);
					obmoff[(size_t)s2 * n + d] = at + mat_at;      // metas behind all payloads
					mat_at += (size_t)(hi - lo) * 4;
				}
			}
		// G: the send buffer of this slot must have left (exchange of two rounds ago)
		if (h.x_recorded[slot])
			HIPCHK(hipStreamWaitEvent(c->stream, h.ev_xdone[slot], 0));
		if (plan.pre[n]) {
			const unsigned g = (unsigned)c->cu_count * 8;
			if (c->nw == 1) hipLaunchKernelGGL(k_sk_gather<SkFmt<1>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			else if (c->nw == 2) hipLaunchKernelGGL(k_sk_gather<SkFmt<2>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			else hipLaunchKernelGGL(k_sk_gather<SkFmt<4>::REC_WORDS>, dim3(g), dim3(256), 0, c->stream, k.p1, k.list1, plan, h.send[slot], h.send_meta[slot], h.recv[slot], h.recv_meta[slot]);
			HIPCHK(hipGetLastError());
		}
		HIPCHK(hipEventRecord(h.ev_gather[slot], c->stream));
		if (t + 1 == S) {                            // pool 1 is free: the next round may scatter while this one travels
			rc = sk_reset_pool1(c);
			if (rc != SDT_OK) return rc;
		}
		// X: on the exchange stream, after the gather and after level 2 has drained this slot's receive buffer
		HIPCHK(hipStreamWaitEvent(cm.xstream, h.ev_gather[slot], 0));
		if (h.l2_recorded[slot])
			HIPCHK(hipStreamWaitEvent(cm.xstream, h.ev_l2[slot], 0));
		{
			// payloads and metas in ONE grouped exchange (one event pair: the time sdt_gpu_comm_stats reports covers both)
			void *const *const sps[2] = {sp.data(), smp.data()}, *const *const rps[2] = {rp.data(), rmp.data()};
			const size_t *const sbs[2] = {sb.data(), smb.data()}, *const rbs[2] = {rb.data(), rmb.data()}, *const obs[2] = {oboff.data(), obmoff.data()};
			rc = cm.exchange(2, sps, sbs, rps, rbs, obs);
		}
		if (rc != SDT_OK) return rc;
		HIPCHK(hipEventRecord(h.ev_xdone[slot], cm.xstream));
		h.x_recorded[slot] = true;
		// B: meanwhile, split + count what the previous exchange brought
		rc = shard_finish_pending(c);
		if (rc != SDT_OK) return rc;
		h.items.swap(cur);
		h.pending = true;
		h.pending_slot = slot;
		h.round++;
	}
	return SDT_OK;
}

extern "C" {
// ---- multi-GPU ----------------------------------------------------------------------------------------------
int sdt_gpu_comm_id(sdt_comm_id *id)
{
	if (!id)
		return fail(SDT_EINVAL, "NULL argument");
	int rc = rccl_load();
	if (rc != SDT_OK)
		return rc;
	static_assert(sizeof(sdt_comm_id) == sizeof(NcclId), "ncclUniqueId is 128 bytes");
	NCCLCHK(g_rccl.GetUniqueId((NcclId *)id));
	return SDT_OK;
}

int sdt_gpu_comm_init(sdt_ctx *c, const sdt_comm_id *id, int rank, int nranks)
{
	if (!c || !id || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	if (c->comm.kind)
		return fail(SDT_ESTATE, "the context already has a communicator");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.open_rccl((const NcclId *)id, rank, nranks);
}

int sdt_gpu_comm_init_shm(sdt_ctx *c, const char *name, int rank, int nranks)
{
	if (!c || !name || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	if (c->comm.kind)
		return fail(SDT_ESTATE, "the context already has a communicator");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.open_shm(name, rank, nranks, true);
}

int sdt_comm_selftest_shm(const char *name, int rank, int nranks, int rounds)
{
	// host-only exercise of the shared-memory transport's control plane (no device): what the CPU tests run with
	// several processes -- all-gather, all-reduce and barriers must agree on every rank, round after round
	Comm cm;
	int rc = cm.open_shm(name, rank, nranks, false);
	if (rc != SDT_OK)
		return rc;
	for (int it = 0; it < rounds && rc == SDT_OK; it++) {
		std::vector<uint32_t> mine(257), all((size_t)257 * nranks);
		for (int i = 0; i < 257; i++) mine[i] = (uint32_t)(rank * 1000003 + it * 7919 + i);
		rc = cm.allgather_host(mine.data(), all.data(), 257 * sizeof(uint32_t));
		for (int r = 0; r < nranks && rc == SDT_OK; r++)
			for (int i = 0; i < 257; i++)
				if (all[(size_t)r * 257 + i] != (uint32_t)(r * 1000003 + it * 7919 + i))
					rc = fail(SDT_EHIP, "all-gather: rank %d got a wrong word from rank %d in round %d", rank, r, it);
		int64_t v[3] = {rank + 1, it, (int64_t)1 << 40};
		if (rc == SDT_OK) rc = cm.allreduce_sum_host(v, 3);
		if (rc == SDT_OK && (v[0] != (int64_t)nranks * (nranks + 1) / 2 || v[1] != (int64_t)it * nranks || v[2] != ((int64_t)nranks << 40)))
			rc = fail(SDT_EHIP, "all-reduce: rank %d got wrong sums in round %d", rank, it);
	}
	cm.close_all();
	return rc;
}

int sdt_gpu_allreduce_i64(sdt_ctx *c, int64_t *vals, int n)
{
	if (!c || !vals || n < 0 || (size_t)n * sizeof(int64_t) > SHM_CTRL_BYTES)
		return fail(SDT_EINVAL, "bad argument");
	HIPCHK(hipSetDevice(c->device));
	return c->comm.allreduce_sum_host(vals, n);
}

int sdt_gpu_comm_stats(sdt_ctx *c, uint64_t *bytes_sent, uint64_t *bytes_recv, double *exchange_ms, uint64_t *exchanges)
{
	if (!c)
		return fail(SDT_EINVAL, "ctx is NULL");
	HIPCHK(hipSetDevice(c->device));
	if (c->comm.xstream)
		HIPCHK(hipStreamSynchronize(c->comm.xstream));
	c->comm.harvest_time();
	if (bytes_sent) *bytes_sent = c->comm.bytes_sent;
	if (bytes_recv) *bytes_recv = c->comm.bytes_recv;
	if (exchange_ms) *exchange_ms = c->comm.exchange_ms;
	if (exchanges) *exchanges = c->comm.exchanges;
	return SDT_OK;
}

int sdt_gpu_shard_ranges(const sdt_ctx *c, uint32_t *first_bucket)
{
	if (!c || !first_bucket)
		return fail(SDT_EINVAL, "NULL argument");
	if (!c->sh.have_ranges)
		return fail(SDT_ESTATE, "no sharded call yet: the bucket ranges are cut on the first one");
	for (int r = 0; r <= c->comm.nranks; r++)
		first_bucket[r] = c->sh.ranges[r];
	return SDT_OK;
}

int sdt_shard_cut_ranges(const uint32_t *mat, int nranks, uint32_t *ranges)
{
	if (!mat || !ranges || nranks < 1 || nranks > SHARD_MAX_RANKS)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks)");
	shard_cut_ranges(mat, nranks, ranges);
	return SDT_OK;
}

int sdt_shard_plan(const uint32_t *mat, int nranks, int me, const uint32_t *ranges, uint32_t recv_chunks, uint32_t t,
                   uint32_t *subrounds, uint32_t *send_begin, uint32_t *send_count, uint32_t *send_at, uint32_t *recv_count,
                   uint32_t *recv_at)
{
	if (!mat || !ranges || !subrounds || !send_begin || !send_count || !send_at || !recv_count || !recv_at)
		return fail(SDT_EINVAL, "NULL argument");
	if (nranks < 1 || nranks > SHARD_MAX_RANKS || me < 0 || me >= nranks || recv_chunks == 0)
		return fail(SDT_EINVAL, "bad argument (1..64 ranks, a receive buffer of at least one chunk)");
	const uint32_t S = shard_subrounds(mat, nranks, ranges, recv_chunks);
	*subrounds = S;
	if (t >= S)
		return fail(SDT_EINVAL, "sub-round %u of %u", t, S);
	ShardRound r;
	shard_round(mat, nranks, me, ranges, t, S, r);
	for (int p = 0; p < nranks; p++) {
		send_begin[p] = r.send_begin[p]; send_count[p] = r.send_count[p]; send_at[p] = r.send_at[p];
		recv_count[p] = r.recv_count[p]; recv_at[p] = r.recv_at[p];
	}
	return SDT_OK;
}
int sdt_kmer_owner(const uint64_t *key_words_msw_first, int K, int nranks)
{
	// owner under EQUAL bucket ranges (what a context uses before its first sharded call has weighed the buckets)
	const int b = sdt_kmer_bucket(key_words_msw_first, K);
	if (b < 0 || nranks < 1)
		return -1;
	return sk_owner_of_bucket((uint32_t)b, nranks);
}
int sdt_gpu_count_reads_sharded(sdt_ctx *c, const void *d_packed_words, uint64_t nwords, const void *d_offsets, uint64_t nreads,
                                uint64_t max_read_len)
{
	(void)nwords;
	if (!c || (nreads && (!d_packed_words || !d_offsets)))
		return fail(SDT_EINVAL, "NULL argument");
	if (c->comm.kind == 0 || c->comm.nranks == 1) {
		if (nreads == 0)
			return SDT_OK;
		return sdt_gpu_count_reads_device(c, d_packed_words, nwords, d_offsets, nreads, max_read_len);
	}
	HIPCHK(hipSetDevice(c->device));
	{ const int rcd = drain_pushes(c, true); if (rcd != SDT_OK) return rcd; }
	Comm &cm = c->comm;
	sdt_ctx::SkState &k = c->sk;
	// agree on the geometry of the call: the longest read anywhere, the rank with the most reads
	std::vector<uint64_t> all((size_t)2 * cm.nranks);
	uint64_t mine[2] = {nreads, nreads ? max_read_len : 0};
	int rc = cm.allgather_host(mine, all.data(), sizeof mine);
	if (rc != SDT_OK) return rc;
	uint64_t maxlen = 0, maxreads = 0;
	for (int r = 0; r < cm.nranks; r++) {
		if (all[2 * r] > maxreads) maxreads = all[2 * r];
		if (all[2 * r + 1] > maxlen) maxlen = all[2 * r + 1];
	}
	if (maxreads == 0 || maxlen < (uint64_t)c->K + 1) {
		c->ord_base += nreads * c->ord_stride;
		return SDT_OK;
	}
	if (maxlen > (uint64_t)SK_MAX_READ_LEN || sk_geo(c->K, maxlen).smem > 160 * 1024)
		return fail(SDT_EINVAL, "reads of %llu bases do not fit the LDS tile of the sharded path", (unsigned long long)maxlen);
	if (c->ord_base + nreads * c->ord_stride >= SK_MAX_READ_ORDINAL)
		return fail(SDT_EINVAL, "read ordinals past 2^34 do not fit a super-k-mer record");
	const uint64_t per_read = maxlen - c->K + 1;
	uint64_t want = maxreads * per_read;
	if (want > (1ULL << 31)) want = 1ULL << 31;       // rounds of at most 2 G k-mers per rank: the exchange overlaps the next round
	if (sdt_test_env("SDT_SHARD_ROUND_KMERS"))             // (tests: many small rounds)
		want = strtoull(sdt_test_env("SDT_SHARD_ROUND_KMERS"), nullptr, 10);
	if (!k.ready || k.cap_kmers < want) {
		if (k.ready && !k.cap_is_max) { HIPCHK(hipStreamSynchronize(c->stream)); sk_free(c); }
		rc = sk_alloc(c, want, per_read);
		if (rc != SDT_OK) return rc;
	}
	rc = shard_alloc(c);
	if (rc != SDT_OK) return rc;
	if (!c->sh.have_ranges) {
		// Ownership.  Minimizer buckets are far from equal (a highly expressed transcript's minimizers are giants), so
		// equal ranges of buckets would leave the ranks unequal work.  Weigh the buckets on a sample -- the first 2^18
		// reads of every rank's slice through the level-1 scatter -- and cut the 256 buckets into contiguous ranges of
		// equal weight.  Every rank computes the same cut from the all-gathered counts; the sample's records are dropped.
		const uint64_t sample = nreads < (1ULL << 18) ? nreads : (1ULL << 18);
		if (sample) {
			rc = sk_scatter_launch(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets, sample, maxlen, c->ord_base, false);
			if (rc != SDT_OK) return rc;
		}
		rc = sk_list1(c);
		if (rc != SDT_OK) return rc;
		std::vector<uint32_t> mat((size_t)cm.nranks * (SK_NB1 + 1));
		rc = cm.allgather_host(k.h_off1, mat.data(), (SK_NB1 + 1) * sizeof(uint32_t));
		if (rc != SDT_OK) return rc;
		shard_cut_ranges(mat.data(), cm.nranks, c->sh.ranges);
		c->sh.have_ranges = true;
		rc = sk_reset_pool1(c);
		if (rc != SDT_OK) return rc;
	}
	// every rank must cut its reads into the same number of rounds
	uint64_t capmine = k.cap_kmers;
	std::vector<uint64_t> caps(cm.nranks);
	rc = cm.allgather_host(&capmine, caps.data(), sizeof capmine);
	if (rc != SDT_OK) return rc;
	uint64_t cap = caps[0];
	for (int r = 1; r < cm.nranks; r++) if (caps[r] < cap) cap = caps[r];
	if (sdt_test_env("SDT_SHARD_ROUND_KMERS") && cap > want) cap = want;
	uint64_t per_round = cap / per_read / SK_TILE_READS * SK_TILE_READS;
	if (per_round < (uint64_t)SK_TILE_READS) per_round = SK_TILE_READS;
	const uint64_t rounds = (maxreads + per_round - 1) / per_round;
	for (uint64_t i = 0; i < rounds; i++) {
		const uint64_t r0 = i * per_round;
		const uint64_t nr = r0 < nreads ? (nreads - r0 < per_round ? nreads - r0 : per_round) : 0;
		if (nr) {
			rc = sk_scatter_launch(c, (const uint32_t *)d_packed_words, (const uint64_t *)d_offsets + r0, nr, maxlen, c->ord_base + r0 * c->ord_stride, false);
			if (rc != SDT_OK) return rc;
			c->sh.kmers_scattered += nr * per_read;
		}
		k.flushing = true;                           // sync_stats must not try to drain the pipeline on its own in here
		rc = sk_flush_sharded(c);
		k.flushing = false;
		if (rc != SDT_OK) return rc;
	}
	k.flushing = true;
	rc = shard_finish_pending(c);
	k.flushing = false;
	c->ord_base += nreads * c->ord_stride;
	return rc;
}

int sdt_gpu_push_reads_sharded(sdt_ctx *c, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads)
{
	if (!c || (nreads && (!packed_words || !offsets)))
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(c->device));
	uint64_t maxlen = 0;
	for (uint64_t i = 0; i < nreads; i++) {
		if (offsets[i + 1] < offsets[i])
			return fail(SDT_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)i);
		if (offsets[i + 1] - offsets[i] > maxlen) maxlen = offsets[i + 1] - offsets[i];
	}
	if (nreads && ((offsets[nreads] + 15) >> 4) + TAIL_PAD > nwords)
		return fail(SDT_EINVAL, "packed_words too short");
	// staging buffers of the single-rank path (slot 0); the call is synchronous with respect to them -- a batch that an earlier
	// asynchronous push left staged there is launched first
	{ const int rcd = drain_pushes(c, true); if (rcd != SDT_OK) return rcd; }
	uint32_t *dw = nullptr;
	uint64_t *dof = nullptr;
	if (nreads) {
		HIPCHK(hipStreamSynchronize(c->stream));
		if (c->cap_words[0] < nwords) {
			if (c->d_words[0]) HIPCHK(hipFree(c->d_words[0]));
			c->d_words[0] = nullptr; c->cap_words[0] = 0;
			HIPCHK(hipMalloc((void **)&c->d_words[0], nwords * sizeof(uint32_t)));
			c->cap_words[0] = nwords;
		}
		if (c->cap_offs[0] < nreads + 1) {
			if (c->d_offs[0]) HIPCHK(hipFree(c->d_offs[0]));
			c->d_offs[0] = nullptr; c->cap_offs[0] = 0;
			HIPCHK(hipMalloc((void **)&c->d_offs[0], (nreads + 1) * sizeof(uint64_t)));
			c->cap_offs[0] = nreads + 1;
		}
		dw = c->d_words[0]; dof = c->d_offs[0];
		HIPCHK(hipMemcpyAsync(dw, packed_words, nwords * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(dof, offsets, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	}
	const int rc = sdt_gpu_count_reads_sharded(c, dw, nwords, dof, nreads, maxlen);
	if (rc == SDT_OK)
		HIPCHK(hipStreamSynchronize(c->stream));     // the staging buffers may be overwritten by the next call
	return rc;
}
} // extern "C"
