/* graph.h -- the k-mer graph on the host after the GPU hashing pass, in the REFERENCE's visiting order.
 *
 * The reference keeps `thrd_num` open-addressing tables (newhash.c) and every later phase walks them
 * "set 0..p-1, slot 0..size-1" while mutating neighbours (cutTipPreGraph.c, node2edge.c), so its results are a
 * function of that slot order.  The slot order in turn is a function of (a) which set a key
 * hashes to (hash_kmer % p, hashFunction.c) and (b) the order in which the distinct keys of a set were first
 * inserted (put_kmerset + encap_kmerset growth/rehash, newhash.c:293-462) -- repeat hits never move entries.
 * graph_build() therefore takes the nodes exported by sdt_gpu_export_nodes with their first-occurrence
 * ordinals, replays only that layout (slot -> node id; no payload moves), and stores the nodes in one flat
 * array in visiting order.  Lookups go through an index of our own (not the reference's probing).
 */
#ifndef SDT_GRAPH_H
#define SDT_GRAPH_H
#include <stdint.h>
#include <stdio.h>
#include <pthread.h>
#include "kw.h"

typedef struct {
	kw_t seq;
	uint32_t l_links;                  /* 4 x 6 bit; edge id once edges are built (node2edge.c:493-519) */
	uint32_t r_links : 24, linear : 1, deleted : 1, checked : 1, single : 1, twin : 2, inEdge : 2;
	uint32_t count;
} gnode_t;

/* (K+1)-mer of an edge of length 1 (two adjacent junction k-mers): the reference keeps these in a second
 * set of tables, KmerSetsPatch (node2edge.c:404-463), looked up by prlRead2edge */
typedef struct {
	kw_t seq;              /* canonical (K+1)-mer */
	uint32_t edge;         /* kmer_t.l_links of the patch node */
	uint8_t twin, used;
} gpatch_t;

#define MO_RW 14                       /* words per junction record of dev_minor_out */

typedef struct graph_s {
	int K, nw, p;
	uint64_t n;
	gnode_t *nodes;                    /* visiting order */
	uint64_t *set_start;               /* p + 1 offsets into nodes[] */
	uint32_t *index;                   /* open addressing: node id + 1, 0 = empty (n < 2^32 - 2) ... */
	uint64_t *index64;                 /* ... or, for larger graphs (and SDT_WIDE_INDEX=1: tests), the same with 64-bit entries;
	                                    * exactly one of the two is set (the reference's sets have ubyte8 sizes, inc/newhash.h:79-88) */
	uint64_t index_mask;
	gpatch_t *patch;                   /* open addressing over canonical (K+1)-mers */
	uint64_t patch_mask, patch_n;
	uint64_t num_ed;                   /* edge_c of kmer2edges: ids handed out, twins included */
	uint8_t *touched;                  /* per node, during a cleaning sweep: written since the dry run (cuttip.c) */
	uint64_t *tlist;                   /* indices marked in `touched` during the current sweep */
	size_t tn, tcap;
	/* device dry runs (optional): when dev_walks is set, the tip passes take their walks from it instead of the
	 * host dry run.  `dirty` lists the nodes written since the device mirror was last brought up to date; the hook
	 * sends them over, and the commit uses the same marks to tell which recorded walks are still what the
	 * reference would walk. */
	uint8_t *dirty;
	uint64_t *dlist;
	size_t dn, dcap;
	/* malloc'ed records of 3 words (node index | info << 56, end index, component label) for the nodes that have a walk,
	 * sorted by (label, node index): a component of the commit is a run of records (cuttip.c) */
	int (*dev_walks)(struct graph_s *g, int thin, int cut_len, uint64_t **records, uint64_t *n_records);   /* 0 = ok */
	/* kmer2edges' port walks from the device: malloc'ed records (17 words each, sdt_gpu_edge_ports) */
	int (*dev_edge_ports)(struct graph_s *g, uint64_t **records, uint64_t *n_records);
	/* removeMinorOut's dry run from the device: malloc'ed records of MO_RW words (node index, 8 neighbours, their 8 occurrence
	 * counts two per word, component label);
	 * the junction records [0, n_junctions) sorted by (label, node index), then the neighbours to cut in any order */
	int (*dev_minor_out)(struct graph_s *g, double threshold, uint64_t **records, uint64_t *n_junctions, uint64_t *n_records);
	/* removeMinorOut's commit on the device as well (sdt_gpu_minor_out_commit_begin / _finish), in two hooks so that the caller's
	 * threads and the device work side by side.  _begin brings the mirror up to date, runs the dry run, starts the commit of the
	 * short components on the device and returns -- malloc'ed, MO_RW words each -- the records of the components the device leaves
	 * alone (too long for one lane): their *n_skipped junction records in order, then, up to *n_skipped_records, the records of the
	 * neighbours they may cut; the caller commits those (they share no node with the others).  _finish waits for the device,
	 * applies the nodes it wrote to nodes[] (graph_apply_written) and returns *off = its kmers off, *linear = the nodes it newly
	 * marked linear.  0 = ok */
	int (*dev_minor_out_commit_begin)(struct graph_s *g, double threshold, uint64_t **skipped, uint64_t *n_skipped, uint64_t *n_skipped_records);
	int (*dev_minor_out_commit_finish)(struct graph_s *g, uint64_t *off, uint64_t *linear);
	/* the whole of kmer2edges from the device (sdt_gpu_build_edges): malloc'ed edge records in id order -- 4 + 2 * *key_words
	 * words each: length | bal_edge << 32, cvg, id, offset into *bases, first and last oriented k-mer (most significant word
	 * first) -- and the edges' bases as letters.  Returns 0, or 2 when a chain does not lead back to the port it was entered
	 * from (the host then builds the edges the sequential way), anything else on failure.  edges_on_device: the interior
	 * nodes were stamped on the device only (their path words wait there for the second read pass) */
	int (*dev_build_edges)(struct graph_s *g, uint64_t **records, int *key_words, uint64_t *n_edges, uint64_t *num_ed, char **bases, uint64_t *n_bases);
	int edges_on_device;
	void *dev_user;
	uint32_t *nb_slot;                 /* per node: 1 + index into nb_pool of its precomputed neighbours, 0 = none */
	uint64_t *nb_pool;                 /* 8 entries per slot: (neighbour index << 1 | smaller) for LEFT 0..3, RIGHT 0..3 */
	uint32_t *nb_cnt;                  /* optional, 8 entries per slot: the occurrence counts of those neighbours */
	pthread_t edge_writer;             /* <prefix>.edge.gz is written in the background (graph_build_edges); graph_edges_join waits */
	int edge_writer_on;
} graph_t;

/* keys: nw words per node, most significant first; r_flags as exported (r_links | linear<<24 | deleted<<25 |
 * single<<27).  Returns NULL after printing a message on failure. */
/* -a n (initKmerSetSize): non-zero changes the initial set size of the 63mer / 127mer variants (prlHashReads.c:404-413) */
extern int graph_init_kmerset_size;
extern int graph_force_wide_index;       /* the host's node index with 64-bit entries whatever the node count (the CLI past its node limit) */
graph_t *graph_build(int K, int nw_variant, int nw_keys, int p, uint64_t n, const uint64_t *keys,
                     const uint32_t *l_links, const uint32_t *r_flags, const uint32_t *count, const uint64_t *first);
/* the same in two steps when the nodes come grouped by set and ordered by first occurrence (sdt_gpu_layout_sorted_keys): the
 * replay alone -- order[v] = index into keys[] of the node at visiting position v -- and, once the device has handed the
 * nodes over in that order (sdt_gpu_layout_apply, sdt_gpu_export_ordered), the graph around them */
void graph_replay_order(int nw_variant, int nw_keys, int p, const uint64_t *keys, const uint64_t *set_start, uint64_t *order);
graph_t *graph_from_ordered(int K, int nw_variant, int nw_keys, int p, uint64_t n, const uint64_t *keys, const uint32_t *l_links,
                            const uint32_t *r_flags, const uint32_t *count, const uint64_t *set_start);
void graph_free(graph_t *g);
/* dirty[dlist[k]] = 0 for the whole list, on all threads (tens of millions of scattered bytes: 90 ms on one thread); dn stays */
void graph_clear_dirty(graph_t *g);
/* free() of up to four large blocks on a detached thread (munmap of gigabytes is not free) */
void graph_free_later(void *a, void *b, void *c, void *d);
/* optional: called by graph_build once the nodes are in visiting order; returns 0 after filling g->index /
 * g->index_mask itself (the GPU host does, from the device's copy of the nodes), non-zero to let the host build it */
extern int (*graph_index_hook)(graph_t *g, void *user);
extern void *graph_index_hook_user;
/* non-zero: the hook reads nothing of nodes[] (the device has the node order already): graph_from_ordered runs it on a thread of its
 * own beside the unpacking of the nodes */
extern int graph_index_hook_early;

/* hash_kmer (hashFunction.c:83-122) for an nw-word variant */
uint64_t ref_hash_kmer(const kw_t *k, int nw);

/* canonical lookup of an ORIENTED k-mer; *smaller = 1 when the oriented word is the stored (smaller) strand.
 * A miss is fatal in the reference (cutTipPreGraph.c:124-140): prints and exit(1). */
gnode_t *graph_find_oriented(graph_t *g, kw_t word, int *smaller);

/* cutTipPreGraph.c */
uint64_t graph_remove_minor_out(graph_t *g, int dd);       /* prints the reference's lines, returns kmers off */
uint64_t graph_remove_single_tips(graph_t *g);
uint64_t graph_remove_minor_tips(graph_t *g);

/* kmer2edges (node2edge.c:46-56): walks every unbranched chain between two non-linear nodes once, writes
 * <prefix>.edge.gz, numbers the edges in visiting order, stamps interior nodes with their edge id.
 * Returns num_ed. */
uint64_t graph_build_edges(graph_t *g, const char *prefix);
/* <prefix>.edge.gz is complete when this returns (its writer runs beside whatever the caller does next) */
void graph_edges_join(graph_t *g);
const gpatch_t *graph_find_patch(const graph_t *g, const kw_t *canon_kplus1);

/* prlRead2edge (prlRead2path.c:817-1335): reads -> edge paths -> arcs -> <prefix>.preArc */
struct arcs;
struct arcs *arcs_new(void);
void arcs_free(struct arcs *A);
void arcs_add_read(graph_t *g, struct arcs *A, const uint8_t *codes, int len, uint64_t ordinal);
int arcs_write(struct arcs *A, const char *prefix);
int arcs_write_arrays(const char *prefix, const uint32_t *from, const uint32_t *to, const uint32_t *mult,
                      const uint64_t *first, uint64_t n);

/* CPU stand-ins for the three device hooks above, made from the host's own dry runs (sdt-graphcheck with
 * SDT_GRAPHCHECK_EMULATE=1, the CPU tests): the commits that sdt-pregraph runs on the device's records run on these */
void graph_emulate_device(graph_t *g);
/* links and flags of n nodes as the device sends them back (l_links; r_links | linear << 24 | deleted << 25): into nodes[], all threads */
void graph_apply_written(graph_t *g, const uint64_t *node, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n);

/* output_pregraph.c:47-81 */
uint64_t graph_write_vertex(graph_t *g, const char *prefix);
extern int graph_vertex_quiet;       /* non-zero: graph_write_vertex does not print its line */
int graph_write_basic(const char *prefix, uint64_t vertices, int K, uint64_t num_ed, int max_read_len);

#endif
