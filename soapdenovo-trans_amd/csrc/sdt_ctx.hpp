// sdt_ctx.hpp -- the device context behind the C ABI (include/sdt_gpu.h) and what the translation units of the library share:
//   sdt_gpu.hip       context life cycle, node table (growth, scans, export / import), the direct pass-1 family, host-buffer pushes
//   sdt_pipeline.hip  the locality pipeline's host side: pools, level-1 / level-2 scatter launches, the count stage's plan + launches
//   sdt_sharded.hip   multi-GPU: the exchange of level-1 chunks between the ranks, communicator glue
//   sdt_pass2.hip     second read pass (prlRead2edge): path words, patch table, k_map_reads, arcs
//   sdt_mapstage.hip  the map stage (prlContig2nodes / prlRead2Ctg)
//   sdt_gpu_graph.hip the graph phases (own view of the context: sdt_internal.hpp GraphView)
// Not part of the ABI: nothing here is visible to a caller of libsdt_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include <new>
#include <thread>
#include <mutex>
#include <atomic>

#include "sdt_internal.hpp"
#include "sdt_superkmer.cuh"

using namespace sdt;
using sdti::fail;
#include "sdt_comm.cuh"
#include "sdt_shard_plan.h"
static_assert(SHARD_NB1 == SK_NB1, "the exchange plan and the pipeline agree about the level-1 buckets");

struct ArcEnt;                 // sdt_map_kernels.cuh

struct EventPair {
	hipEvent_t a, b;
	uint64_t kmers;
	int stage;               // SDT_STAGE_*
};

struct sdt_ctx {
	int device = 0;
	int K = 0;
	int nw = 1;
	uint64_t slots = 0;
	void *d_ent = nullptr;
	uint32_t *d_aux = nullptr;
	uint64_t *d_first = nullptr;       // SDT_FLAG_TRACK_FIRST
	uint64_t ord_base = 0, ord_stride = 1;
	Stats *d_stats = nullptr;
	Stats *h_stats = nullptr;          // pinned
	unsigned long long *d_hist = nullptr;
	hipStream_t stream = nullptr, copy_stream = nullptr;
	bool own_stream = true;
	// host-batch staging (double buffered)
	// A ring of NSTAGE device buffers.  A pushed batch is COPIED at once (copy stream) and queued; its kernels are launched
	// from the queue.  The one thing that blocks the host for long is a flush of the locality pipeline (two host syncs for
	// the chunk lists: ~30 ms per 2^31 k-mers), and while the host is blocked nobody feeds the copy engine -- so a launch
	// that needs a flush is put off until STAGE_AHEAD copies are queued behind it: the copies then run while the host waits
	// (without this the PCIe-inclusive rate was compute + copy, not max(compute, copy): 50 vs 76 G k-mers/s resident).
	static constexpr int NSTAGE = 48, STAGE_AHEAD = 32;
	uint32_t *d_words[NSTAGE] = {};
	uint64_t *d_offs[NSTAGE] = {};
	uint64_t cap_words[NSTAGE] = {}, cap_offs[NSTAGE] = {};
	hipEvent_t buf_free[NSTAGE] = {}, copied[NSTAGE] = {};
	int next_buf = 0;
	struct Staged { const uint32_t *dw; uint64_t *dof; uint64_t nreads, maxlen, ord_base, ord_stride; int slot; uint64_t fixed_len; };
	std::vector<Staged> staged;        // copied (or being copied), not yet launched: [staged_head, size)
	size_t staged_head = 0;
	bool draining = false;
	uint64_t push_ord_base = 0, push_ord_stride = 1;      // ordinals of the next PUSHED batch (ord_base / ord_stride: of the next LAUNCHED one)
	uint64_t push_ticket = 0;          // pushes issued so far: ticket t's host buffers are free once copied[(t - 1) % NSTAGE] has passed
	uint64_t expect_kmers = 0;         // sdt_gpu_hint_total_kmers
	// bookkeeping for growth: upper bound of distinct nodes without syncing
	uint64_t distinct_known = 0;       // as of the last sync
	uint64_t kmers_known = 0;          // occurrences counted as of the last sync (new nodes per occurrence: bound of the next launch)
	uint64_t kmers_since_sync = 0;     // launched since then (an upper bound of the new nodes they may bring)
	uint64_t hard_since_sync = 0;      // k-mers launched since then by the locality pipeline, whatever its own bound said
	uint64_t kmers_total_host = 0;
	uint64_t kmers_offered = 0;        // upper bound of the k-mers handed to pass 1 since the last reset (picks the kernel family)
	uint32_t flags = 0;
	// locality pipeline (sdt_superkmer.cuh): chunk pools of the two scatter levels, chunk lists, pending work
	struct SkState {
		bool ready = false;
		uint64_t cap_kmers = 0;            // k-mers the pools are sized for (one batch)
		bool cap_is_max = false;           // the device has no room for larger pools: do not try again
		uint64_t pending_kmers = 0;        // scattered into pool 1, not yet counted
		SkPool p1 = {nullptr, nullptr, nullptr, 0}, p2 = {nullptr, nullptr, nullptr, 0};
		unsigned long long *cursors = nullptr;   // [wgs][SK_NB1] open chunks of the level-1 scatter
		unsigned long long *blk = nullptr;       // [wgs] block of chunk ids each workgroup is handing out
		uint32_t wgs = 0;
		uint32_t *cnt1 = nullptr, *off1 = nullptr, *fill1 = nullptr, *list1 = nullptr;
		uint32_t *cnt2 = nullptr, *off2 = nullptr, *fill2 = nullptr, *list2 = nullptr;
		unsigned long long *kmers2 = nullptr, *kpre2 = nullptr;
		SkItem *items = nullptr;
		uint32_t items_cap = 0;
		uint4 *citems = nullptr, *h_citems = nullptr;       // work items of k_sk_count: [c0, c1) in list2 + their final buckets (device / pinned; sdt_count_plan.h)
		uint32_t citems_cap = 0;
		uint32_t *next_item = nullptr;                      // one counter per k_sk_count launch
		uint32_t *h_off1 = nullptr, *h_off2 = nullptr;      // pinned
		unsigned long long *h_kpre2 = nullptr;              // pinned
		SkItem *h_items = nullptr;                          // pinned
		bool flushing = false;
		// statistics of the last flush (sdt_gpu_pipeline_stats)
		uint64_t st_records = 0, st_chunks1 = 0, st_chunks2 = 0, st_flushes = 0;
		uint32_t stream_flushes = 0;   // flushes since the last reset (sk_batch_limit)
		uint64_t l2_in_total = 0;      // k-mers that entered the count stage (sum of the level-2 bucket sizes): Stats.sk_counted must match
		bool exchanged = false;        // records left for / came from other ranks: Stats.sk_emitted is not this rank's input
	} sk;
	// multi-GPU (sdt_comm.cuh): communicator + double-buffered send / receive chunk buffers of the exchange
	Comm comm;
	struct Shard {
		uint64_t *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};       // chunk payloads
		uint32_t *send_meta[2] = {nullptr, nullptr}, *recv_meta[2] = {nullptr, nullptr};
		uint32_t *iota = nullptr;                                                    // identity chunk list of a receive buffer
		uint32_t send_chunks = 0, recv_chunks = 0;
		hipEvent_t ev_gather[2] = {nullptr, nullptr}, ev_xdone[2] = {nullptr, nullptr}, ev_l2[2] = {nullptr, nullptr};
		bool x_recorded[2] = {false, false}, l2_recorded[2] = {false, false};
		uint64_t round = 0;
		// what the last exchange delivered and sk_split has not consumed yet
		bool pending = false;
		int pending_slot = 0;
		uint32_t pending_items = 0;
		std::vector<SkItem> items;
		uint64_t kmers_scattered = 0;
		// which rank owns which level-1 buckets: contiguous ranges [ranges[r], ranges[r + 1]), balanced by the bucket
		// weights of a sample of the first call's reads (the same on every rank: the weights are all-gathered)
		bool have_ranges = false;
		uint32_t ranges[65] = {0};
	} sh;
	// second pass (prlRead2edge): reads kept from pass 1, path words, patch table, arcs
	struct KeptBatch { uint32_t *d_words; uint64_t *d_offs; uint64_t nwords, nreads, ord_base, ord_stride, maxlen; };
	std::vector<KeptBatch> kept;
	// kept batches live in a few large slabs (two hipMallocs per 32 MiB batch were thousands of synchronous calls on
	// the ingest path): bump allocation, everything is released together
	struct KeepSlab { uint8_t *p; size_t size, used; };
	std::vector<KeepSlab> keep_slabs;
	void *d_patch = nullptr;
	uint64_t patch_slots = 0;
	ArcEnt *d_arcs = nullptr;
	uint64_t arc_slots = 0;
	bool paths_loaded = false;
	uint64_t *d_idx = nullptr;         // slot -> index of the node in the host's visiting order (sdt_gpu_layout_apply / sdt_gpu_set_node_index)
	sdti::GraphExt *gx = nullptr;      // graph phases (sdt_gpu_graph.hip)
	uint64_t idx_slots = 0, idx_n = 0;
	// map stage (SDT_FLAG_CONTIG_INDEX): contig ordinal -> id, contig_array, staging for sdt_gpu_align_reads
	uint32_t *d_ctg_ids = nullptr;
	uint64_t ctg_ord = 0, ctg_ids_cap = 0;
	uint32_t *d_ctg_len = nullptr, *d_ctg_twin = nullptr;
	uint64_t num_ctg = 0;
	void *ab[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // words, offsets, align_len, read_info, hits
	size_t ab_cap[5] = {0, 0, 0, 0, 0};
	unsigned long long *d_hit_cursor = nullptr;
	bool index_final = false;            // k_finalize_contig_index has run: look-ups only from here on
	// timing
	std::vector<EventPair> ev;
	size_t ev_used = 0;
	int cu_count = 256;
};

constexpr double MAX_LOAD = 0.70;
// the flat table: what the direct kernel family counts into (and grows)
template <int NW> inline Table<NW> flat_of(const sdt_ctx *c)
{
	Table<NW> t;
	t.ent = (Entry<NW> *)c->d_ent;
	t.aux = c->d_aux;
	t.fslots = c->slots;
	t.first = c->d_first;
	return t;
}

// the node table as every stage after pass 1 sees it
template <int NW> inline Table<NW> table_of(const sdt_ctx *c) { return flat_of<NW>(c); }
inline uint64_t view_slots(const sdt_ctx *c) { return c->slots; }

inline size_t entry_bytes(int nw) { return nw == 1 ? sizeof(Entry<1>) : nw == 2 ? sizeof(Entry<2>) : sizeof(Entry<4>); }

inline int scan_grid(const sdt_ctx *c, uint64_t items) { return sdti::scan_grid(c->cu_count, items); }


inline int env_int(const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; }
inline int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- sdt_gpu.hip ----
uint64_t flat_slots_for(uint64_t nodes);
void *keep_alloc(sdt_ctx *c, size_t bytes);
void keep_release(sdt_ctx *c);
int launch_clear(sdt_ctx *c, void *ent, uint32_t *aux, uint64_t *first, uint64_t slots);
int alloc_table(sdt_ctx *c, uint64_t slots, void **ent, uint32_t **aux, uint64_t **first);
int sync_stats(sdt_ctx *c);
int grow_table(sdt_ctx *c, uint64_t need_nodes);
int ensure_room(sdt_ctx *c, uint64_t incoming);
EventPair *next_event(sdt_ctx *c);
size_t tile_smem_bytes(int max_tile_words);
int tile_words_for(uint64_t max_read_len);
int drain_pushes(sdt_ctx *c, bool force);      // batches that were pushed but not launched yet are launched (force: all of them)
// ---- sdt_pipeline.hip ----
int sk_flush(sdt_ctx *c);
void sk_free(sdt_ctx *c);
// ---- sdt_sharded.hip ----
void shard_free(sdt_ctx *c);
int sk_flush_sharded(sdt_ctx *c);
