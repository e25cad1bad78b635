/*
 * sdt_gpu.h -- C ABI of libsdt_gpu.so: the MI355X (gfx950) implementation of SOAPdenovo-Trans's
 * `pregraph` hashing path (read chopping -> canonical k-mer -> hash insert/count -> -d filter ->
 * linear marking + k-mer frequency histogram -> node export).
 *
 * The reference has no FFI layer: the path sits behind the C functions that call_pregraph() calls
 * (inc/extfunc.h:82,156-163) with state in globals.  Each entry point below names the reference
 * function / call site it replaces (paths relative to /root/reference/src).  A host written in C
 * (soapdenovo-trans_amd/csrc/host/sdt_pregraph.c), Python/ctypes (soapdenovo-trans_amd/__init__.py)
 * or the reference's own prlHashReads.c binds exactly these symbols.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative SDT_E* code and
 * leaves a message retrievable by sdt_gpu_last_error(); no exceptions cross the boundary; one host
 * thread per context; a context owns one GPU.  There is NO CPU fallback: without a usable gfx950
 * device sdt_gpu_init fails with SDT_ENODEV.
 *
 * Packed reads ("2-bit stream"): all reads of a batch concatenated, 2 bits per base with the
 * reference's coding A=0 C=1 T=2 G=3 (inc/def.h:39-42), 16 bases per little-endian uint32 word, FIRST
 * base in the MOST significant bit pair (so a k-mer is a funnel shift of consecutive words and equals
 * the reference's Kmer value, inc/def.h:45-59).  read i occupies bases [offsets[i], offsets[i+1]).
 * The word array must be readable for 4 words past the last base (pad with zeros).
 */
#ifndef SDT_GPU_H
#define SDT_GPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDT_ABI_VERSION 8

enum {
	SDT_OK       = 0,
	SDT_EINVAL   = -1,   /* bad argument */
	SDT_ENODEV   = -2,   /* no gfx950 device / HIP runtime error at init */
	SDT_ENOMEM   = -3,   /* device or host allocation failed */
	SDT_EHIP     = -4,   /* HIP runtime error (message in sdt_gpu_last_error) */
	SDT_EFULL    = -5,   /* node table cannot grow any further */
	SDT_ESTATE   = -6,   /* call out of order (e.g. export before finish); also: the pipeline's conservation check failed
	                      * (k-mers cut into records != k-mers counted, sdt_gpu_finish_count) -- never a silent loss */
	SDT_ELIMIT   = -7    /* a data-dependent limit of a device algorithm was passed (sdt_gpu_layout_on_device: rounds of a growth,
	                      * depth of an eviction chain); nothing was changed: the caller takes the other path */
};

typedef struct sdt_ctx sdt_ctx;

/* sdt_gpu_init flags: pick the pass-1 kernel family.  Both give identical tables.  Default (neither flag): the locality
 * pipeline wherever its geometry applies (reads of K+1 .. ~550 bases, one rank), the direct kernel otherwise. */
#define SDT_FLAG_DIRECT    1u   /* always one device atomic per k-mer occurrence (k_count_reads) */
#define SDT_FLAG_PARTITION 2u   /* locality pipeline (csrc/sdt_superkmer.cuh): minimizer buckets of super-k-mers, counted in LDS,
                                 * one merge per distinct key and batch */
/* Track, per node, the ordinal of its first occurrence in the read stream: (read ordinal << 16) | position.
 * The reference's table layout -- hence the visiting order of its cutting passes, the order of *.vertex and the
 * edge ids -- is a function of exactly this order; the host replays it (csrc/host/graph). */
#define SDT_FLAG_TRACK_FIRST 4u
/* Keep every pushed batch of packed reads resident in HBM so that the second pass over the reads
 * (prlRead2edge, sdt_gpu_map_reads) needs no re-parse: 0.25 B/base, 7.5 GB for 200 M x 150 bp. */
#define SDT_FLAG_KEEP_READS 8u
/* map stage: the table indexes the k-mers of the contigs (sdt_gpu_index_contigs; implies TRACK_FIRST) */
#define SDT_FLAG_CONTIG_INDEX 16u
/* (bits 32 and 64 were SDT_FLAG_FLAT_MERGE / SDT_FLAG_NODE_LOG until ABI 7: the count stage of the locality pipeline merges every
 * generation of its LDS table into the node table, the only form since ABI 8; the bits are ignored) */

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* Replaces the allocation half of prlRead2HashTable (prlHashReads.c:355,402-423: createFilter +
 * init_kmerset x thrd_num).  K = overlaplen after call_pregraph's clamp (pregraph.c:38-59): odd, 13..127.
 * est_distinct sizes the device node table (it grows by rebuild when needed, the analogue of
 * encap_kmerset, newhash.c:293-409); 0 = default.  device = HIP device ordinal. */
int sdt_gpu_init(sdt_ctx **ctx, int device, int K, uint64_t est_distinct, uint32_t flags);
int sdt_gpu_destroy(sdt_ctx *ctx);                 /* free_Sets (pregraph.c:107) */
const char *sdt_gpu_last_error(void);
int sdt_gpu_abi_version(void);

/* forget all nodes, keep allocations (a fresh prlRead2HashTable run on the same context) */
int sdt_gpu_reset(sdt_ctx *ctx);

/* ---- pass 1: chop + insert/count ------------------------------------------------------------- */

/* Replaces one `sendWorkSignal(2); sendWorkSignal(1);` pair (prlHashReads.c:523-526,600-606,615-620):
 * chopKmer4read over every read of the batch (:164-310) and put_kmerset of every record
 * (newhash.c:411-462).  Host buffers; the call stages them to the device asynchronously (double
 * buffered) and returns once the buffers may be reused.  Reads shorter than K+1 are skipped (:592). */
int sdt_gpu_push_reads(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords,
                       const uint64_t *offsets, uint64_t nreads);

/* The same without the wait: the call returns as soon as the copies and kernels are enqueued (a ring of 48 device staging
 * buffers, up to 32 batches staged ahead of their kernels; the host blocks only when the ring is full).  The caller's buffers must
 * stay untouched until sdt_gpu_push_wait(ctx, *ticket) has returned -- a host that parses into a ring of its own waits for
 * the ticket of the buffer it is about to refill, not for every push (prlHashReads.c:493-620 double-buffers the same way:
 * one buffer is parsed while the threads work on the other).  Pinned host memory keeps the copy asynchronous.
 * hint_total_kmers: the caller expects this many k-mers in all, so the first small batch already takes the locality pipeline
 * (cleared by sdt_gpu_reset). */
int sdt_gpu_push_reads_async(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads,
                             uint64_t *ticket);
/* a batch whose reads all have read_len bases (read i starts at base i * read_len): no offsets cross PCIe, the device makes them */
int sdt_gpu_push_reads_fixed_async(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, uint64_t nreads, uint64_t read_len,
                                   uint64_t *ticket);
int sdt_gpu_push_wait(sdt_ctx *ctx, uint64_t ticket);
/* pinned host memory for the buffers of asynchronous pushes (the copy engine reads it directly; a push from pageable memory is
 * staged by the runtime at a fraction of the link and blocks the caller): stands where the reference mallocs its two read
 * buffers (prlHashReads.c:430-470).  NULL when no memory can be pinned.  Callable from any host thread. */
void *sdt_gpu_host_alloc(size_t bytes);
void sdt_gpu_host_free(void *p);
int sdt_gpu_hint_total_kmers(sdt_ctx *ctx, uint64_t kmers);

/* Read ordinals of the NEXT batch: read i of it gets ordinal base + i*stride; afterwards the base advances by
 * nreads*stride.  Only needed with SDT_FLAG_TRACK_FIRST when the stream is not consumed file after file: the
 * reference interleaves paired files read1, read2, read1, ... (prlHashReads.c:493-567) = stride 2, base 0 / 1. */
int sdt_gpu_set_read_ordinal(sdt_ctx *ctx, uint64_t base, uint64_t stride);

/* Same, for a batch that is already resident in device memory (device pointers; the bench and the
 * multi-GPU driver use this).  Asynchronous on the context's stream.  max_read_len bounds the longest
 * read of the batch (the reference's maxReadLen, prlHashReads.c:358-366); it sizes the LDS tile. */
int sdt_gpu_count_reads_device(sdt_ctx *ctx, const void *d_packed_words, uint64_t nwords,
                               const void *d_offsets, uint64_t nreads, uint64_t max_read_len);

/* Drain: all pushed batches are in the table on return (end of the read loop, prlHashReads.c:615-623).
 * Outputs (may be NULL): k-mer occurrences processed ("kmer in reads", :662) and distinct nodes
 * ("nodes allocated" = sum of count_kmerset, :655-662).  SDT_EFULL if an insert ever found no slot; SDT_ESTATE if the
 * k-mers that went into the locality pipeline's records are not the k-mers that came out of its count stage. */
int sdt_gpu_finish_count(sdt_ctx *ctx, uint64_t *kmers_processed, uint64_t *nodes);

/* ---- multi-GPU, bucket sharding (the product path) ---------------------------------------------------------
 * The reference partitions records over threads by hash_kmer % thrd_num (prlHashReads.c:79-88).  Across the GPUs of a
 * node every rank owns a contiguous range of the 256 level-1 minimizer buckets (csrc/sdt_superkmer.cuh): all
 * occurrences of a canonical k-mer -- either strand, any read -- fall into one bucket, so they meet on one rank.
 * What travels is the super-k-mer record (~3 B per k-mer occurrence instead of a 16-B (key, meta) record), in
 * level-1 chunks: one grouped ncclSend / ncclRecv per peer and round over xGMI, on a stream of its own, while the
 * next round's reads are chopped and the previous round's records are split and counted.
 *   comm_id / comm_init      one process per GPU; rank 0 makes the id (ncclGetUniqueId) and hands it to the others
 *   comm_init_shm            same protocol over POSIX shared memory + host staging: validation where several ranks
 *                            share one GPU (RCCL refuses that), never for a reported number
 *   count_reads_sharded      COLLECTIVE: every rank calls it with ITS slice of the reads (device buffers as in
 *                            sdt_gpu_count_reads_device; nreads may be 0).  On return the k-mers of all slices are in
 *                            the tables of their owners (asynchronously: sdt_gpu_finish_count drains).
 *   push_reads_sharded       the same for host buffers
 *   allreduce_i64            COLLECTIVE sum, for counters and the 257 kmerFreq bins (freqStat sums per-thread bins,
 *                            prlHashReads.c:1004-1014)
 *   comm_stats               bytes this rank sent / received in exchanges and the time they took on the exchange stream
 *   shard_ranges / sdt_kmer_bucket   who owns what: rank r owns the level-1 buckets [first_bucket[r], first_bucket[r + 1])
 *                            (nranks + 1 entries), cut on the FIRST sharded call so that the ranks' bucket weights -- taken
 *                            from a sample of every rank's reads -- are equal; sdt_kmer_bucket is the host copy of the device's
 *                            bucket function (canonical k-mer -> 0..255).  sdt_kmer_owner: the owner under EQUAL ranges. */
typedef struct { unsigned char bytes[128]; } sdt_comm_id;
int sdt_gpu_comm_id(sdt_comm_id *id);
int sdt_gpu_comm_init(sdt_ctx *ctx, const sdt_comm_id *id, int rank, int nranks);
int sdt_gpu_comm_init_shm(sdt_ctx *ctx, const char *name, int rank, int nranks);
int sdt_gpu_count_reads_sharded(sdt_ctx *ctx, const void *d_packed_words, uint64_t nwords, const void *d_offsets,
                                uint64_t nreads, uint64_t max_read_len);
int sdt_gpu_push_reads_sharded(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets,
                               uint64_t nreads);
int sdt_gpu_allreduce_i64(sdt_ctx *ctx, int64_t *vals, int n);
int sdt_gpu_comm_stats(sdt_ctx *ctx, uint64_t *bytes_sent, uint64_t *bytes_recv, double *exchange_ms, uint64_t *exchanges);
int sdt_gpu_shard_ranges(const sdt_ctx *ctx, uint32_t *first_bucket);
int sdt_kmer_bucket(const uint64_t *key_words_msw_first, int K);

/* the FINAL minimizer bucket (0 .. 2^18 - 1) of a canonical k-mer: the unit the count stage of the locality pipeline works on (one
 * workgroup counts all occurrences of a bucket's keys in LDS) */
int sdt_kmer_final_bucket(const uint64_t *key_words_msw_first, int K);
/* the node table as it stands: info[1] slots, [2] nodes as of the last look at the device's counters; the other words are 0
 * (they described the node log of ABI 7) */
int sdt_gpu_table_info(sdt_ctx *ctx, uint64_t info[8]);
int sdt_kmer_owner(const uint64_t *key_words_msw_first, int K, int nranks);
/* After pass 1 the order-dependent graph phases (cutTipPreGraph.c, node2edge.c) run on ONE host over ALL nodes: rank 0
 * takes the other ranks' exported shards (sdt_gpu_export_nodes arrays; keys are disjoint by construction) into its own
 * table with import_nodes -- its device mirror then answers the dry runs and the second read pass for the whole
 * graph -- and keeps every read of the run resident with keep_reads (like SDT_FLAG_KEEP_READS, minus the counting;
 * ordinals from sdt_gpu_set_read_ordinal as for a push). */
int sdt_gpu_import_nodes(sdt_ctx *ctx, const uint64_t *keys, const uint32_t *l_links, const uint32_t *r_flags,
                         const uint32_t *count, const uint64_t *first, uint64_t n);
int sdt_gpu_keep_reads(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets, uint64_t nreads);
/* host-only self test of the shared-memory transport's control plane (no device needed; the CPU tests run it with
 * several processes) */
int sdt_comm_selftest_shm(const char *name, int rank, int nranks, int rounds);

/* The exchange plan as pure host functions of the count matrix (csrc/sdt_shard_plan.h; no device, no communicator): what
 * sdt_gpu_count_reads_sharded computes after its all-gather, exported so that the protocol can be driven -- and checked --
 * with any transport (tests/test_multirank_gloo.py: two gloo ranks on CPU).
 *   mat[r * 257 + b]   first position of level-1 bucket b in rank r's chunk list (b = 256: the list's length)
 *   cut_ranges         ranges[0..nranks]: rank d owns buckets [ranges[d], ranges[d + 1]), equal weights
 *   plan               sub-round t of the exchange as rank `me` sees it (*subrounds: how many there are -- the same on every
 *                      rank): for every peer p the piece [send_begin[p], + send_count[p]) of my chunk list, its place
 *                      send_at[p] in my send buffer (p == me: in my receive buffer), and the run recv_count[s] at recv_at[s]
 *                      of my receive buffer that arrives from rank s.  Arrays of nranks entries. */
int sdt_shard_cut_ranges(const uint32_t *mat, int nranks, uint32_t *ranges);
int sdt_shard_plan(const uint32_t *mat, int nranks, int me, const uint32_t *ranges, uint32_t recv_chunks, uint32_t t,
                   uint32_t *subrounds, uint32_t *send_begin, uint32_t *send_count, uint32_t *send_at, uint32_t *recv_count,
                   uint32_t *recv_at);

/* The work items and launches of the count stage as a pure host function of the level-2 chunk lists (csrc/sdt_count_plan.h; no
 * device): what the library computes between the level-2 scatter and k_sk_count, callable so that it can be tested on a CPU.
 *   off2[f], kpre2[f]   first chunk / first k-mer of final bucket f in the chunk list (f = nbuckets: the totals)
 *   items               4 words per work item: the run [c0, c1) of the list, top bit of c1 = the item holds whole buckets only;
 *                       its first and its last final bucket.
 *                       Buckets of <= 64 chunks share an item with their neighbours -- inside one level-1 bucket and a span of 64
 *                       final buckets --, buckets of > 1024 chunks are cut into pieces.
 *   first_item[l], launch_kmers[l]   first item and k-mers of launch l (first_item[*nlaunches] = *nitems); a launch is cut
 *                       between buckets at `limit` k-mers (`first_limit` for the first).
 * Stands where the reference hands a batch of k-mers to its threads (prlHashReads.c:312-336, sendWorkSignal). */
int sdt_sk_plan_count_items(const uint32_t *off2, const uint64_t *kpre2, uint32_t nbuckets, uint64_t first_limit, uint64_t limit,
                            uint32_t max_launches, uint32_t *items, uint32_t items_cap, uint32_t *first_item, uint64_t *launch_kmers,
                            uint32_t launches_cap, uint32_t *nitems, uint32_t *nlaunches);

/* ---- table scans ------------------------------------------------------------------------------ */

/* deLowCov / thread_delow (prlHashReads.c:844-909), `-d d`: zero links with 0 < v <= d, mark nodes
 * left without links deleted.  *removed = "%lld kmer removed". */
int sdt_gpu_delow(sdt_ctx *ctx, int d, uint64_t *removed);

/* Mark1in1outNode / thread_mark (prlHashReads.c:911-992): set `linear`, fill the 257-bin histogram
 * that freqStat (:994-1023) prints bins 1..255 of.  *linear = "%lld linear nodes". */
int sdt_gpu_mark_and_hist(sdt_ctx *ctx, int64_t hist[257], uint64_t *linear);

/* ---- hand the node table back to host graph phases (cutTipPreGraph.c, node2edge.c) ------------
 * Copies every node to caller-owned host arrays of capacity max_nodes (order unspecified):
 *   keys     : key_words() uint64 per node, MOST significant word first (the reference Kmer struct order)
 *   l_links  : kmer_t.l_links (4 x 6-bit, inc/newhash.h:38-43)
 *   r_flags  : the second 32-bit word of kmer_t (inc/newhash.h:69-75): r_links:24 | linear<<24 |
 *              deleted<<25 | checked<<26 | single<<27 | twin<<28 | inEdge<<30
 *   count    : kmer_t.count
 *   first    : first-occurrence ordinal (needs SDT_FLAG_TRACK_FIRST)
 * Any array may be NULL.  *n receives the node count. */
int sdt_gpu_export_nodes(sdt_ctx *ctx, uint64_t *keys, uint32_t *l_links, uint32_t *r_flags,
                         uint32_t *count, uint64_t *first, uint64_t max_nodes, uint64_t *n);

/* --gpus N, second read pass on every rank: the final graph as that pass needs it -- every node's key and path word -- out of the
 * table of the rank that built the edges (after sdt_gpu_load_paths), and into the table of a rank that holds its own share of the
 * reads (sdt_gpu_keep_reads / SDT_FLAG_KEEP_READS): its shard of pass 1 makes way, the kept reads stay; patch table and edge count
 * as in sdt_gpu_load_paths.  Then sdt_gpu_map_reads + sdt_gpu_export_arcs on every rank, and the arcs of all ranks add up
 * (prlRead2path.c:415-430 counts arcs per thread and adds them the same way). */
int sdt_gpu_export_paths(sdt_ctx *ctx, uint64_t *keys, uint64_t *path_words, uint64_t max_nodes, uint64_t *n);
int sdt_gpu_import_paths(sdt_ctx *ctx, const uint64_t *keys, const uint64_t *path_words, uint64_t n, const uint64_t *patch_keys,
                         const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed);
/* A rank whose shard has been handed over (sdt_gpu_export_nodes) and that now waits for the graph: its node table, the
 * first-occurrence ordinals and the pools of the locality pipeline go back to the device; the reads kept for the second pass stay.
 * The context then holds no nodes until sdt_gpu_import_paths gives it the graph.  (The reference frees its sets only at the very
 * end, pregraph.c:107-108: one process, one address space.  Here the waiting ranks may share a device with rank 0's graph phases.) */
int sdt_gpu_release_table(sdt_ctx *ctx);
/* ---- pass 2: reads -> edge paths -> arcs (prlRead2edge, prlRead2path.c:817-1335) --------------------------
 * After the host graph phases (minor-out, tip cutting, kmer2edges) every node gets one path word
 *     bit 0 skip = deleted || (linear && !inEdge) (:650) | bit 1 linear | bits 2..3 twin | bits 32..63 l_links = edge id
 * and the (K+1)-mers of length-1 edges (KmerSetsPatch, node2edge.c:404-463) come as patch_keys (key_words()
 * words each, most significant first) with patch_info = edge id | twin << 32.  load_paths overwrites the
 * nodes' counters with the path words (export the table first); keys == NULL: path_words[i] belongs to node i of
 * sdt_gpu_set_node_index / sdt_gpu_layout_apply (no keys to send, no look-ups); path_words == NULL as well: the path words
 * sdt_gpu_build_edges left on the device.  map_reads then replays parse1read (:617-789),
 * search1kmerPlus (:575-615) and the arc counting (:190-241,415-430) over the kept reads; export_arcs returns
 * every arc with its multiplicity and the ordinal of its first appearance ((read ordinal << 16) | item index):
 * per from-edge the reference prints arcs most-recent-first-appearance first (:427-428,472-496) -- the arrays come in that order
 * (from ascending, first appearance descending) since round 5. */
int sdt_gpu_load_paths(sdt_ctx *ctx, const uint64_t *keys, const uint64_t *path_words, uint64_t n,
                       const uint64_t *patch_keys, const uint64_t *patch_info, uint64_t npatch, uint64_t num_ed);
int sdt_gpu_map_reads(sdt_ctx *ctx, uint64_t *reads_processed, uint64_t *arcs);
int sdt_gpu_export_arcs(sdt_ctx *ctx, uint32_t *from, uint32_t *to, uint32_t *mult, uint64_t *first,
                        uint64_t max_arcs, uint64_t *n);

/* ---- graph-cleaning dry runs (cutTipPreGraph.c) on the device mirror of the host graph ------------------
 * The passes after kmerFreq are order-dependent: the host commits their writes in the reference's
 * order.  What each sweep needs before that is read-only -- the walk from every dead end to the node it runs
 * into (clipTipFromNode, cutTipPreGraph.c:43-281) -- and that is table look-ups: the device table, kept equal to
 * the host graph, answers them for all nodes at once.
 *   set_node_index: keys in the host's visiting order (index i = position in `keys`); results are indexed by it.
 *   update_nodes:   links / linear / deleted of the nodes the host wrote since the last call (l_links, r_flags as
 *                   in export_nodes).
 *   tip_walks:      for every node i: end_idx[i] = index of the node the walk from i stops at, ~0 when there is
 *                   nothing to decide (not a dead end, chain longer than cut_len, deleted, linear; thin != 0:
 *                   removeSingleTips' rule, only `single` nodes start or continue a walk); info[i] = ch | sm << 2 |
 *                   thin_stop << 3: the base by which the end node sees the chain, the strand it was reached on,
 *                   and (thin) whether the walk stopped at a linear node that is not single (:163-166). */
int sdt_gpu_set_node_index(sdt_ctx *ctx, const uint64_t *keys, uint64_t n);
int sdt_gpu_update_nodes(sdt_ctx *ctx, const uint64_t *keys, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n);
int sdt_gpu_tip_walks(sdt_ctx *ctx, int thin, int cut_len, uint64_t *end_idx, uint8_t *info, uint64_t n);
/*   tip_walks_compact: the same walks, only for the nodes that have one, in no particular order: records of 2 words,
 *                   [0] = node index | info << 56, [1] = end index.  SDT_EFULL: *n_records says how many there are. */
int sdt_gpu_tip_walks_compact(sdt_ctx *ctx, int thin, int cut_len, uint64_t *records, uint64_t max_records,
                              uint64_t *n_records);
/*   minor_out_dry:  removeMinorOut's read-only part (cutTipPreGraph.c:1012-1076): every junction whose ratio test
 *                   (count / largest count on that side < threshold = dd / 100.0, clipKmerFromNode :591-1010)
 *                   would cut at least one neighbour on the graph as it is now, and who the neighbours of those
 *                   junctions and of the neighbours to cut are.  records: 9 words each -- node index, then
 *                   (neighbour index << 1 | smaller) or ~0 for the four left and the four right links;
 *                   [0, n_junctions) are the junctions, [n_junctions, n_records) the neighbours to cut that are not
 *                   junctions themselves.  SDT_EFULL when records[] is too small: *n_records says what is needed. */
int sdt_gpu_minor_out_dry(sdt_ctx *ctx, double threshold, uint64_t *records, uint64_t max_records,
                          uint64_t *n_junctions, uint64_t *n_records);
/*   build_host_index: the host's k-mer -> node look-up table for the ordered commits (csrc/host/graph/graph.c:
 *                   open addressing over index_slots = 2^m >= 2n 32-bit words, home slot = mix_key(4-word k-mer) &
 *                   (slots-1), linear probing, value = node index + 1, 0 = empty), filled by the device from its node index. */
int sdt_gpu_build_host_index(sdt_ctx *ctx, uint32_t *index, uint64_t index_slots);
/*   edge_ports:     kmer2edges' walks (node2edge.c:46-191): for every node that is neither linear nor deleted one
 *                   record of 17 words -- node index, then for each of its 8 ports (right links 0..3 on the stored
 *                   strand, left links 0..3 on the reverse strand) the index of the first non-linear node the chain
 *                   of linear nodes leads to (~0: no link) and  length | far_port << 32 | bal_edge << 40  (the port
 *                   the chain arrives through; bal_edge = 0 when the chain is its own reverse complement,
 *                   check_iden_kmerList :563-588).  SDT_EFULL when records[] is too small (*n_records = needed). */
int sdt_gpu_edge_ports(sdt_ctx *ctx, uint64_t *records, uint64_t max_records, uint64_t *n_records);

/* ---- the reference's visiting order, and the dry runs labelled for commits that run side by side ----------------------
 * Every phase after kmerFreq walks the reference's tables "set 0..p-1, slot 0..size-1" (cutTipPreGraph.c:351-366,385-408,
 * 1049-1072, node2edge.c:46-56), and a node's place there is a function of hash_kmer(key) % p and of the order in which the
 * distinct keys of its set first occurred (put_kmerset / encap_kmerset, newhash.c:293-462).  The device knows both:
 *   layout_sorted_keys: sorts the nodes by (set, first-occurrence ordinal) -- set = hash_kmer (hashFunction.c:83-122) over the
 *                   bytes of the nw_variant-word Kmer of the emulated binary, % p -- and returns the KEYS in that order
 *                   (key_words() words each) with set_start[0..p]; keys == NULL: only *n.  Ends pass 1 (its pools are freed).
 *   layout_apply:   order[v] = rank (index into that key array) of the node at visiting position v, from the host's replay of
 *                   the probing (csrc/host/graph/graph.c: graph_replay_order).  Numbers the nodes: everything below that
 *                   speaks of a node index means v.  Replaces sdt_gpu_set_node_index (no keys cross the link).
 *   export_ordered: the nodes in visiting order, arrays as sdt_gpu_export_nodes (any may be NULL).
 *   update_nodes_by_index: sdt_gpu_update_nodes with node indices instead of keys.
 *   tip_walks_labelled / minor_out_labelled: the dry runs of sdt_gpu_tip_walks_compact / sdt_gpu_minor_out_dry with one more
 *                   word per record, the COMPONENT of the record's node: a visit of the ordered commit reads and writes only
 *                   its own node and nodes of the same component, so components commit side by side, each in the reference's
 *                   order (csrc/host/graph/cuttip.c).  Components = union-find over node indices on the device --
 *                   removeSingleTips (thin): tip + end node of every walk; removeMinorTips: non-linear nodes joined by chains
 *                   of <= cut_len linear nodes; removeMinorOut: every record's node + its eight neighbours.  Records:
 *                   walks 3 words (node | info << 56, end, label), all sorted by (label, node); junctions 14 words (node,
 *                   8 neighbours, their 8 occurrence counts two per word, label), the first *n_junctions sorted by
 *                   (label, node), then the neighbours to cut.
 *                   The records stay on the device until fetch_records copies them (nwords = records x words, exactly). */
int sdt_gpu_layout_sorted_keys(sdt_ctx *ctx, int p, int nw_variant, uint64_t *keys, uint64_t max_nodes, uint64_t *set_start, uint64_t *n);
int sdt_gpu_layout_apply(sdt_ctx *ctx, const uint64_t *order, uint64_t n);
/*   layout_on_device: layout_sorted_keys + the replay of put_kmerset / encap_kmerset (newhash.c:293-462) + layout_apply in one call,
 *                   nothing but set_start[0..p] crosses the link.  Between two growths a set is laid out by priority insertion
 *                   (first come first served in first-occurrence order, built in any order); a growth -- the in-place rehash of
 *                   :359-406, where an entry that gives way is carried on at once -- as a fixed point of insertion times, ten to
 *                   twenty rounds of priority insertion (csrc/sdt_graph_kernels.cuh; tools/replay_fixed_point.c checks the
 *                   formulation against the sequential emulation).  small_init != 0: the sets of the 63mer / 127mer variants
 *                   start at 3 slots (`-a`, prlHashReads.c:404-413).  SDT_EINVAL when a limit is passed (2^32 nodes, a set's table
 *                   of 2^32 slots, the packed table word): use the two-step form with the host's replay then.
 *                   Round 5: SDT_ELIMIT for those limits and for a growth that does not settle; the rounds are incremental (only the
 *                   stretch from a changed entry's home to the end of its cluster is laid out again); on success the first-occurrence
 *                   ordinals are dropped (8 bytes per table slot: layout_sorted_keys / export_nodes(first) return SDT_ESTATE after). */
int sdt_gpu_layout_on_device(sdt_ctx *ctx, int p, int nw_variant, int small_init, uint64_t *set_start, uint64_t *n);
int sdt_gpu_export_ordered(sdt_ctx *ctx, uint64_t *keys, uint32_t *l_links, uint32_t *r_flags, uint32_t *count, uint64_t n);
int sdt_gpu_update_nodes_by_index(sdt_ctx *ctx, const uint64_t *node, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n);
int sdt_gpu_tip_walks_labelled(sdt_ctx *ctx, int thin, int cut_len, uint64_t *n_records);
int sdt_gpu_minor_out_labelled(sdt_ctx *ctx, double threshold, uint64_t *n_junctions, uint64_t *n_records);
int sdt_gpu_fetch_records(sdt_ctx *ctx, uint64_t *dst, uint64_t nwords);
/*   minor_out_commit: removeMinorOut's COMMIT (cutTipPreGraph.c:591-1010: the ratio test on the live links, `deleted`, the
 *                   neighbours' links cleared and their `linear` re-derived) on the records minor_out_labelled left on the device,
 *                   one lane per component, the visits of a component in the reference's order; then thread_mark's re-marking
 *                   over the nodes it wrote (:911-967).  *off = "kmers off", *linear = nodes newly marked linear, *n_written =
 *                   nodes whose links or flags changed -- fetch_written copies them out as (index, l_links, r_links | linear << 24 |
 *                   deleted << 25), the form update_nodes_by_index takes.  Components of more than max_component visits (*largest
 *                   = the largest there is) are left alone: one lane is no match for a host thread on a long chain of dependent
 *                   accesses.  fetch_skipped hands their records over (*n_skipped_records of 14 words: first their *n_skipped junction
 *                   records in order, then the records of the neighbours they may cut); the caller commits them (they touch no node
 *                   the device wrote) and sends what it wrote with update_nodes_by_index. */
int sdt_gpu_minor_out_commit(sdt_ctx *ctx, double threshold, uint64_t max_component, uint64_t *largest, uint64_t *off, uint64_t *linear, uint64_t *n_written,
                             uint64_t *n_skipped, uint64_t *n_skipped_records);
int sdt_gpu_fetch_skipped(sdt_ctx *ctx, uint64_t *dst, uint64_t n_records);
/*   minor_out_commit in two halves, so that the caller can commit the long components while the device walks the short ones:
 *                   _begin finds the components, gathers the records of the long ones (fetch_skipped may be called right after it)
 *                   and LAUNCHES the visits; _finish waits for them, re-marks and lists the written nodes (fetch_written). */
int sdt_gpu_minor_out_commit_begin(sdt_ctx *ctx, double threshold, uint64_t max_component, uint64_t *largest, uint64_t *n_skipped, uint64_t *n_skipped_records);
int sdt_gpu_minor_out_commit_finish(sdt_ctx *ctx, uint64_t *off, uint64_t *linear, uint64_t *n_written);
int sdt_gpu_fetch_written(sdt_ctx *ctx, uint64_t *node, uint32_t *l_links, uint32_t *r_flags, uint64_t n);
/* kmer2edges (node2edge.c:46-561) on the device mirror, after sdt_gpu_layout_apply: every chain of linear nodes between two
 * nodes that are neither linear nor deleted is one edge; it belongs to the first of its two (node, port) ends in visiting order
 * (ports: right links 0..3 on the stored strand, then left links 0..3 on the other), ids are handed out in that order (an edge
 * that is not its own reverse complement takes two), the interior nodes are stamped with id and twin (merge_linearV2, :351-561)
 * -- as PATH WORDS: sdt_gpu_load_paths(ctx, NULL, NULL, n, ...) then takes them from the device.  *n_edges records wait for
 * sdt_gpu_fetch_records (4 + 2 * key_words() words each, in id order: length | bal_edge << 32, cvg, id, offset of the edge's
 * bases, first and last oriented k-mer) and *n_bases letters for sdt_gpu_fetch_edge_bases (the last base of nodes 1..length of
 * every edge); *num_ed = ids handed out (EDGEs of *.preGraphBasic).  SDT_ESTATE with "does not lead back" in the message: a
 * chain is not symmetric -- nothing was stamped, build the edges sequentially (node2edge.c's own order). */
int sdt_gpu_build_edges(sdt_ctx *ctx, uint64_t *n_edges, uint64_t *num_ed, uint64_t *n_bases);
int sdt_gpu_fetch_edge_bases(sdt_ctx *ctx, char *dst, uint64_t nbytes);

/* ---- `map` stage: prlContig2nodes (prlHashCtg.c:287-425) and prlRead2Ctg (prlRead2Ctg.c:656-894) -------
 * A context created with SDT_FLAG_CONTIG_INDEX holds the k-mers of the contigs:
 *   index_contigs: contigs packed like reads (2 bit / base, offsets in bases, 4 pad words), ids[i] = the id the
 *                  reference takes from the record name (getID, prlHashCtg.c:276-285) -- the caller has applied
 *                  the length cut (:343-350).  May be called repeatedly; contig order = call order, array order.
 *                  The first occurrence of a k-mer (contig order, then position) owns contig id / position /
 *                  strand, every further one marks it deleted (singleKmer :110-139).
 *                  sdt_gpu_finish_count then reports "kmer in reads" and "nodes allocated" (:397).  The first
 *                  align call freezes the index (nodes are rewritten into their look-up form): SDT_ESTATE after.
 *   set_contig_table: contig_array[0..num_ctg] of basicContigInfo (prlRead2Ctg.c:610-648): length, and
 *                  twin[i] = getTwinCtg(i) (attachPEinfo.c:479-482).
 *   align_reads:   chopKmer4read + searchKmer + parse1read (prlRead2Ctg.c:129-353) for a batch of reads.
 *                  align_len: per-read ALIGNLEN (the value of the global when the read's batch is parsed,
 *                  :774-791), or NULL and align_len_all for every read.
 *                  read_info[r] = more_start (40 bits) | nhits << 40 | best << 48 | footprint << 56 | overflow << 57;
 *                  nhits = count_Contig (0: ctgIdArray[t] = 0).  ctg2read[t][0] = hits[r]; ctg2read[t][m], m >= 1,
 *                  = hits[more_start + m - 1] (the tail of hits[] past the first nreads entries, in the reference's
 *                  order); best = index m of the hit that sets ctgIdArray / posArray / orienArray (posArray =
 *                  contig_offset - read_offset + 1); overflow: more than 20 candidate contigs -- the reference
 *                  overruns pos_temp[20] there; such a read is reported unmapped.
 *                  max_hits >= nreads; *nhits = entries of hits[] in use (nreads + all further hits); SDT_EFULL
 *                  when hits[] is too small: *nhits says how many the batch needs.
 *   align_reads_device: the same on buffers already in device memory (outputs too). */
typedef struct {
	uint32_t contig;             /* READSET.contigID */
	int32_t contig_offset;       /* READSET.contigOffset */
	uint32_t read_offset;        /* READSET.readOffset (1-based k-mer index) */
	uint32_t align_len_orien;    /* READSET.alignLength | (orien == '-' ? 1u << 31 : 0) */
} sdt_hit;
int sdt_gpu_index_contigs(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets,
                          const uint32_t *ids, uint64_t ncontigs);
int sdt_gpu_set_contig_table(sdt_ctx *ctx, const uint32_t *length, const uint32_t *twin, uint64_t num_ctg);
int sdt_gpu_align_reads(sdt_ctx *ctx, const uint32_t *packed_words, uint64_t nwords, const uint64_t *offsets,
                        uint64_t nreads, const int32_t *align_len, int align_len_all, uint64_t *read_info,
                        sdt_hit *hits, uint64_t max_hits, uint64_t *nhits);
int sdt_gpu_align_reads_device(sdt_ctx *ctx, const void *d_packed_words, const void *d_offsets, uint64_t nreads,
                               uint64_t max_read_len, const void *d_align_len, int align_len_all, void *d_read_info,
                               void *d_hits, uint64_t max_hits, uint64_t *nhits);

/* ---- introspection / measurement --------------------------------------------------------------- */
int sdt_gpu_key_words(const sdt_ctx *ctx);         /* 1 (K<=31), 2 (K<=63), 4 (K<=127) */
uint64_t sdt_gpu_table_slots(const sdt_ctx *ctx);
void *sdt_gpu_stream(const sdt_ctx *ctx);          /* hipStream_t the kernels are launched on */
/* launch on a caller-owned hipStream_t instead (NULL = back to a private stream); used by the multi-GPU
 * driver so that RCCL collectives and our kernels are ordered by one stream */
int sdt_gpu_set_stream(sdt_ctx *ctx, void *hip_stream);
/* HIP-event timing of the dominant (chop+insert) kernel accumulated since the last call with reset!=0:
 * total milliseconds, number of launches, k-mer occurrences those launches processed. */
int sdt_gpu_kernel_time(sdt_ctx *ctx, int reset, double *ms, uint64_t *launches, uint64_t *kmers);
/* Where the time of pass 1 went, by stage (HIP events on the context's stream, summed since the last
 * sdt_gpu_kernel_time(reset != 0)), and the locality pipeline's counters since the last reset:
 *   counters[0] LDS nodes merged into the table   [1] k-mers that took the direct path out of a full LDS table
 *           [2] k-mers that took it because the chunk pool was exhausted   [3] early flushes of a full LDS table
 *           [4] / [5] level-1 / level-2 chunks of the last batch   [6] batches counted   [7] k-mers per batch the pools hold
 *           [8..11] k_sk_count: 100 MHz clock ticks summed over workgroups in set-up / tile fill / counting / merging
 *           [12..15] k_sk_scatter_reads: the same for tile staging / window minima / run starts / emission */
#define SDT_STAGE_DIRECT      0   /* k_count_reads / k_insert_records: one atomic per occurrence */
#define SDT_STAGE_SK_SCATTER  1   /* k_sk_scatter_reads: chop + minimizers + level-1 scatter */
#define SDT_STAGE_SK_SPLIT    2   /* chunk lists + k_sk_scatter_records (level 2) */
#define SDT_STAGE_SK_COUNT    3   /* k_sk_count: LDS counting + merges */
#define SDT_STAGE_SK_FOLD     4   /* (ABI 7: the fold of the node log; always 0 now) */
#define SDT_NSTAGES           5
#define SDT_NCOUNTERS         20   /* [16] distinct records of the count stage's tiles, [17] records (level-2), [18] k-mers of the distinct records, [19] reserved */
int sdt_gpu_stage_times(sdt_ctx *ctx, double ms[SDT_NSTAGES], uint64_t counters[SDT_NCOUNTERS]);
/* the device table's slot hash of a canonical key (host-callable, identical to the device function) */
uint64_t sdt_owner_hash(const uint64_t *key_words_msw_first, int nwords);

#ifdef __cplusplus
}
#endif
#endif
