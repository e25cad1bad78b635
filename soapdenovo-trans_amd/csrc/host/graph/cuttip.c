/* cuttip.c -- minor-branch removal and tip cutting on the host graph, in the reference's visiting order.
 *
 * What each pass must reproduce (cutTipPreGraph.c; decision rules also in SURVEY.md 9.7):
 *   removeMinorOut (:1012-1076)   one sweep; at a node with more than one predecessor (then: successor) every
 *                                 linked neighbour whose occurrence count is < dd% of the largest such count
 *                                 is marked deleted and unlinked from all of ITS neighbours (:591-1010)
 *   removeSingleTips (:339-370)   one sweep, THIN: dead ends whose chain is made of single-occurrence k-mers
 *   removeMinorTips (:372-437)    per set, sweeps until nothing changes: a dead-end chain of <= 2K nodes is cut
 *                                 when its link into the junction is not the strongest on that side (:43-337)
 *   Mark1in1outNode (:1193-1229)  after each pass: not-deleted, not-yet-linear nodes with 1 in / 1 out -> linear
 * All of them mutate neighbours while sweeping, so the sweep order (graph.h) is part of the result.
 */
#include "../../sdt_knobs.h"
#include "graph.h"
#include "par.h"
#include "big.h"
#include <stdlib.h>
#include <string.h>

/* ---- speculate in parallel, commit in order ---------------------------------------------------------------
 * The passes are order dependent only through the few nodes they WRITE: a clip writes its tip and the node the
 * chain runs into; a minor-branch cut writes the cut neighbour and that neighbour's neighbours.  Everything
 * else a visit reads is immutable during the pass (occurrence counts, `single`, and every node that is linear
 * at the start of a sweep: walks only ever stop at -- and therefore write -- non-linear nodes, and nothing in a
 * sweep clears `linear` on a node a walk could pass through).  So each sweep is run as
 *   1. a read-only dry run of every visit on the untouched graph, in parallel: "would this visit write
 *      anything?" and "which node did its walk end at?";
 *   2. the reference's sweep, in order, executing a visit for real only if its node has been written since the
 *      dry run, or the dry run said it writes, or the node its walk ended at has been written; every other
 *      visit would read exactly what the dry run read and do nothing, so it is skipped.
 * Writes happen only in step 2, in the reference's order, by the same code as before: results are identical,
 * and the expensive part (the walks of the ~95 % of visits that change nothing) is parallel. */
#include <time.h>
static double cut_now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
#define SUBPHASE(name) do { if (sdt_env("SDT_TIMING")) { double t_ = cut_now_ms(); fprintf(stderr, "[cuttip]   %-26s %9.1f ms\n", name, t_ - t_sub); t_sub = t_; } } while (0)
#define NO_NODE (~(uint64_t)0)
typedef struct { int would_write; uint64_t end; } dry_t;

/* parallel commits (removeMinorOut by components): a thread only ever touches nodes of its own component, so the
 * per-node marks need no synchronisation; the shared list is replaced by one list per thread, merged afterwards */
static __thread struct { uint64_t *v; size_t n, cap; int on; } tl_dirty;
struct tip_comp;
static __thread struct tip_comp *tl_comp;          /* the component of walks this thread is committing (tip passes) */
static void comp_first_touch(struct tip_comp *C, uint64_t i);

static inline void touch(graph_t *g, const gnode_t *n)
{
	const uint64_t i = (uint64_t)(n - g->nodes);
	if (tl_dirty.on) {
		if (g->dirty && !g->dirty[i]) {
			g->dirty[i] = 1;
			if (tl_dirty.n == tl_dirty.cap) {
				tl_dirty.cap = tl_dirty.cap ? tl_dirty.cap * 2 : 4096;
				tl_dirty.v = (uint64_t *)realloc(tl_dirty.v, tl_dirty.cap * sizeof(uint64_t));
			}
			tl_dirty.v[tl_dirty.n++] = i;
			if (tl_comp) comp_first_touch(tl_comp, i);
		}
		return;
	}
	if (g->dirty && !g->dirty[i]) {
		g->dirty[i] = 1;
		if (g->dn == g->dcap) {
			g->dcap = g->dcap ? g->dcap * 2 : 4096;
			g->dlist = (uint64_t *)realloc(g->dlist, g->dcap * sizeof(uint64_t));
		}
		g->dlist[g->dn++] = i;
	}
	if (!g->touched || g->touched[i]) return;
	g->touched[i] = 1;
	if (g->tn == g->tcap) {
		g->tcap = g->tcap ? g->tcap * 2 : 4096;
		g->tlist = (uint64_t *)realloc(g->tlist, g->tcap * sizeof(uint64_t));
	}
	g->tlist[g->tn++] = i;                     /* so that a sweep can un-mark what it marked without a full clear */
}

enum { LEFT = 0, RIGHT = 1 };

static inline unsigned link_of(const gnode_t *n, int side, unsigned b)
{
	return ((side == LEFT ? n->l_links : n->r_links) >> (6 * b)) & 63u;
}
static inline void drop_link(gnode_t *n, int side, unsigned b)
{
	if (side == LEFT) n->l_links &= ~(63u << (6 * b));
	else n->r_links = n->r_links & ~(63u << (6 * b));
}
static inline int degree(const gnode_t *n, int side)
{
	int d = 0;
	for (unsigned b = 0; b < 4; b++) d += link_of(n, side, b) != 0;
	return d;
}
static inline int one_in_one_out(const gnode_t *n) { return degree(n, LEFT) == 1 && degree(n, RIGHT) == 1; }

/* forget the link of `n` (reached in orientation `smaller`) towards the k-mer that precedes / follows it in
 * the walk and whose adjacent base is ch (dislink2prevUncertain / dislink2nextUncertain, newhash.c:540-562) */
static inline void unlink_prev(gnode_t *n, unsigned ch, int smaller) { if (smaller) drop_link(n, LEFT, ch); else drop_link(n, RIGHT, ch ^ 2u); }
static inline void unlink_next(gnode_t *n, unsigned ch, int smaller) { if (smaller) drop_link(n, RIGHT, ch); else drop_link(n, LEFT, ch ^ 2u); }

/* Mark1in1outNode (cutTipPreGraph.c:1121-1229): a scan; the few nodes it marks are reported to touch() afterwards
 * (the device mirror has to hear about them) */
typedef struct { graph_t *g; uint64_t *list[64]; size_t n[64], cap[64]; } ml_ctx;

static void mark_linear_part(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	ml_ctx *M = (ml_ctx *)vc;
	graph_t *g = M->g;
	for (uint64_t i = lo; i < hi; i++) {
		gnode_t *n = &g->nodes[i];
		if (n->deleted || n->linear) continue;
		if (!one_in_one_out(n)) continue;
		n->linear = 1;
		if (M->n[tid] == M->cap[tid]) {
			M->cap[tid] = M->cap[tid] ? M->cap[tid] * 2 : 1024;
			M->list[tid] = (uint64_t *)realloc(M->list[tid], M->cap[tid] * sizeof(uint64_t));
		}
		M->list[tid][M->n[tid]++] = i;
	}
}

static void mark_linear_dirty(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	ml_ctx *M = (ml_ctx *)vc;
	graph_t *g = M->g;
	size_t c = 0;
	for (uint64_t k = lo; k < hi; k++) {
		if (k + 8 < hi) __builtin_prefetch(&g->nodes[g->dlist[k + 8]], 1);
		gnode_t *n = &g->nodes[g->dlist[k]];
		if (n->deleted || n->linear || !one_in_one_out(n)) continue;
		n->linear = 1;
		c++;
	}
	M->n[tid] += c;
}

static uint64_t mark_linear(graph_t *g)
{
	uint64_t c = 0;
	if (g->dirty) {
		/* with the device mirror every write of a pass is on the dirty list (touch()), and a node can only have become
		 * 1-in-1-out by being written: the list is all there is to look at (the marked nodes are on it already: the mirror
		 * hears about them with the rest) */
		ml_ctx M;
		memset(&M, 0, sizeof M);
		M.g = g;
		par_for(0, g->dn, 1 << 12, mark_linear_dirty, &M);
		for (int t = 0; t < 64; t++) c += M.n[t];
	} else {
		ml_ctx M;
		memset(&M, 0, sizeof M);
		M.g = g;
		par_for(0, g->n, 1 << 16, mark_linear_part, &M);
		for (int t = 0; t < 64; t++) {
			for (size_t k = 0; k < M.n[t]; k++) touch(g, &g->nodes[M.list[t][k]]);
			c += M.n[t];
			free(M.list[t]);
		}
	}
	printf("%d thread created for cutTipPreGraph\n", g->p);
	printf("%llu linear nodes\n", (unsigned long long)c);
	return c;
}

/* neighbour of the CANONICAL k-mer of n across its `side` link with base b */
static gnode_t *neighbour(graph_t *g, const gnode_t *n, int side, unsigned b, int *smaller)
{
	/* who the neighbour IS depends only on n's k-mer and b, never on the state of the graph: a pass may look
	 * the neighbours of the nodes it is going to need up front, in parallel, and the ordered commit reads them here */
	if (g->nb_slot) {
		const uint32_t s = g->nb_slot[n - g->nodes];
		if (s) {
			const uint64_t v = g->nb_pool[(uint64_t)(s - 1) * 8 + (uint64_t)side * 4 + b];
			if (v != NO_NODE) {
				*smaller = (int)(v & 1);
				return &g->nodes[v >> 1];
			}
		}
	}
	return graph_find_oriented(g, side == LEFT ? kw_prev(n->seq, b, g->K) : kw_next(n->seq, b, g->K), smaller);
}

static void isolate(graph_t *g, gnode_t *q)
{
	const kw_t qs = q->seq;
	int sm;
	q->deleted = 1;
	touch(g, q);
	for (unsigned b = 0; b < 4; b++)
		if (link_of(q, LEFT, b)) {
			gnode_t *x = neighbour(g, q, LEFT, b, &sm);
			unlink_next(x, kw_last(&qs), sm);
			x->linear = one_in_one_out(x);
			touch(g, x);
		}
	for (unsigned b = 0; b < 4; b++)
		if (link_of(q, RIGHT, b)) {
			gnode_t *y = neighbour(g, q, RIGHT, b, &sm);
			unlink_prev(y, kw_first(&qs, g->K), sm);
			y->linear = one_in_one_out(y);
			touch(g, y);
		}
}

/* occurrence count of the neighbour across link (side, b): from the dry run's records when they carry it (counts never change) */
static inline int neighbour_count(graph_t *g, const gnode_t *n, int side, unsigned b)
{
	if (g->nb_cnt && g->nb_slot) {
		const uint32_t s = g->nb_slot[n - g->nodes];
		if (s && g->nb_pool[(uint64_t)(s - 1) * 8 + (uint64_t)side * 4 + b] != NO_NODE)
			return (int)g->nb_cnt[(uint64_t)(s - 1) * 8 + (uint64_t)side * 4 + b];
	}
	int sm;
	return (int)neighbour(g, n, side, b, &sm)->count;
}

static void prune_side(graph_t *g, gnode_t *n, int side, double threshold, uint64_t *off, dry_t *dry)
{
	int sm, best = 0;
	for (unsigned b = 0; b < 4; b++)
		if (link_of(n, side, b)) {
			const int c = neighbour_count(g, n, side, b);
			if (c > best) best = c;
		}
	if (!best) return;
	for (unsigned b = 0; b < 4; b++) {
		if (!link_of(n, side, b)) continue;                   /* live: an earlier cut may have removed it */
		const int c = neighbour_count(g, n, side, b);
		if (c && (double)c / best < threshold) {
			if (dry) { dry->would_write = 1; return; }
			(*off)++;
			isolate(g, neighbour(g, n, side, b, &sm));
		}
	}
}

static void visit_minor_out(graph_t *g, gnode_t *n, double threshold, uint64_t *off, dry_t *dry)
{
	if (n->linear || n->deleted) return;
	const int in = degree(n, LEFT), out = degree(n, RIGHT);            /* both sampled before any cut (:616-617) */
	if (in <= 1 && out <= 1) return;
	if (in > 1) prune_side(g, n, LEFT, threshold, off, dry);
	if (dry && dry->would_write) return;
	if (out > 1) prune_side(g, n, RIGHT, threshold, off, dry);
}

/* removeMinorOut, parallel part: for every node that is a junction when the pass starts, look its neighbours up
 * (needed for the counts) and flag the neighbours that the ratio test would cut on the untouched graph -- links
 * only ever disappear during the pass, so the largest count on a side can only go down and the live cut set is a
 * subset of this one; then look up the neighbours of the flagged nodes as well (isolate() needs them). */
typedef struct { graph_t *g; double threshold; uint8_t *need, *writes; volatile uint32_t cursor; uint32_t cap; } mo_ctx;

static void fill_slot(graph_t *g, uint64_t i, uint32_t slot)
{
	const gnode_t *n = &g->nodes[i];
	uint64_t *e = &g->nb_pool[(uint64_t)(slot - 1) * 8];
	for (int side = 0; side < 2; side++)
		for (unsigned b = 0; b < 4; b++) {
			e[side * 4 + b] = NO_NODE;
			if (!link_of(n, side, b)) continue;
			int sm;
			const gnode_t *x = graph_find_oriented(g, side == LEFT ? kw_prev(n->seq, b, g->K) : kw_next(n->seq, b, g->K), &sm);
			e[side * 4 + b] = ((uint64_t)(x - g->nodes) << 1) | (uint64_t)sm;
		}
}

static void mo_junctions(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	mo_ctx *c = (mo_ctx *)vc;
	graph_t *g = c->g;
	for (uint64_t i = lo; i < hi; i++) {
		const gnode_t *n = &g->nodes[i];
		if (n->linear || n->deleted) continue;
		const int in = degree(n, LEFT), out = degree(n, RIGHT);
		if (in <= 1 && out <= 1) continue;
		/* neighbours + the ratio test on the untouched graph */
		uint64_t e[8];
		int any = 0;
		for (int side = 0; side < 2; side++)
			for (unsigned b = 0; b < 4; b++) {
				e[side * 4 + b] = NO_NODE;
				if (!link_of(n, side, b)) continue;
				int sm;
				const gnode_t *x = graph_find_oriented(g, side == LEFT ? kw_prev(n->seq, b, g->K) : kw_next(n->seq, b, g->K), &sm);
				e[side * 4 + b] = ((uint64_t)(x - g->nodes) << 1) | (uint64_t)sm;
			}
		for (int side = 0; side < 2; side++) {
			if ((side == LEFT ? in : out) <= 1) continue;
			int best = 0;
			for (unsigned b = 0; b < 4; b++)
				if (e[side * 4 + b] != NO_NODE) {
					const int cnt = (int)g->nodes[e[side * 4 + b] >> 1].count;
					if (cnt > best) best = cnt;
				}
			if (!best) continue;
			for (unsigned b = 0; b < 4; b++)
				if (e[side * 4 + b] != NO_NODE) {
					const int cnt = (int)g->nodes[e[side * 4 + b] >> 1].count;
					if (cnt && (double)cnt / best < c->threshold) { c->need[e[side * 4 + b] >> 1] = 1; any = 1; }
				}
		}
		if (!any) continue;                       /* the visit would change nothing unless somebody writes this node first */
		c->writes[i] = 1;
		const uint32_t slot = __sync_add_and_fetch(&c->cursor, 1);
		if (slot > c->cap) continue;              /* pool full: the commit simply looks these up itself */
		memcpy(&g->nb_pool[(uint64_t)(slot - 1) * 8], e, sizeof e);
		g->nb_slot[i] = slot;
	}
}

static void mo_candidates(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	mo_ctx *c = (mo_ctx *)vc;
	graph_t *g = c->g;
	for (uint64_t i = lo; i < hi; i++) {
		if (!c->need[i] || g->nb_slot[i]) continue;
		const uint32_t slot = __sync_add_and_fetch(&c->cursor, 1);
		if (slot > c->cap) continue;
		fill_slot(g, i, slot);
		g->nb_slot[i] = slot;
	}
}

static void count_junctions(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	mo_ctx *c = (mo_ctx *)vc;
	uint32_t k = 0;
	for (uint64_t i = lo; i < hi; i++) {
		const gnode_t *n = &c->g->nodes[i];
		if (n->linear || n->deleted) continue;
		if (degree(n, LEFT) > 1 || degree(n, RIGHT) > 1) k++;
	}
	__sync_fetch_and_add(&c->cursor, k);
}

/* ---- removeMinorOut, commit by components -------------------------------------------------------------------
 * A visit at junction i reads and writes only  T(i) = {i} + N(i) + N(q) for the neighbours q it may cut  (isolate(q)
 * unlinks q from its neighbours), all known from the dry run's neighbour tables.  Visits whose T() are disjoint
 * commute, so: union the visits that share a node (lock-free union-find, in parallel), then run every component's
 * visits in the reference's order on one thread, components side by side. */
typedef struct {
	graph_t *g;
	const uint64_t *ex;
	uint64_t nexec;
	uint32_t *owner;          /* per node: a visit that touches it, ~0 = none yet */
	uint32_t *parent;         /* union-find over visits */
	double threshold;
	uint64_t *order;          /* visits grouped by component, ascending inside each */
	uint64_t *cstart;         /* ncomp + 1 */
	uint64_t ncomp;
	volatile uint64_t off;
	uint64_t **tl;            /* merged after the run */
	size_t *tln;
	volatile int ntl;
} cc_ctx;

static inline uint32_t cc_find(uint32_t *parent, uint32_t v)
{
	for (;;) {
		const uint32_t p = __atomic_load_n(&parent[v], __ATOMIC_RELAXED);
		if (p == v) return v;
		const uint32_t gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
		if (gp != p) __sync_bool_compare_and_swap(&parent[v], p, gp);   /* path halving */
		v = p;
	}
}

static inline void cc_union(uint32_t *parent, uint32_t a, uint32_t b)
{
	for (;;) {
		a = cc_find(parent, a);
		b = cc_find(parent, b);
		if (a == b) return;
		if (a < b) { const uint32_t t = a; a = b; b = t; }               /* the larger root goes under the smaller */
		if (__sync_bool_compare_and_swap(&parent[a], a, b)) return;
	}
}

static inline void cc_claim(cc_ctx *C, uint32_t v, uint64_t node)
{
	const uint32_t old = __sync_val_compare_and_swap(&C->owner[node], 0xFFFFFFFFu, v);
	if (old != 0xFFFFFFFFu && old != v) cc_union(C->parent, v, old);
}

static void cc_clear(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	memset((uint32_t *)vc + lo, 0xFF, (size_t)(hi - lo) * sizeof(uint32_t));
}

static void cc_link(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	cc_ctx *C = (cc_ctx *)vc;
	graph_t *g = C->g;
	for (uint64_t v = lo; v < hi; v++) {
		const uint64_t i = C->ex[v];
		cc_claim(C, (uint32_t)v, i);
		const uint32_t sl = g->nb_slot[i];
		if (!sl) continue;                                                 /* cannot happen when the tables are complete */
		const uint64_t *e = &g->nb_pool[(uint64_t)(sl - 1) * 8];
		for (int k = 0; k < 8; k++) {
			if (e[k] == NO_NODE) continue;
			const uint64_t q = e[k] >> 1;
			cc_claim(C, (uint32_t)v, q);
			const uint32_t s2 = g->nb_slot[q];                              /* q may be cut only if the dry run gave it a table */
			if (!s2) continue;
			const uint64_t *e2 = &g->nb_pool[(uint64_t)(s2 - 1) * 8];
			for (int k2 = 0; k2 < 8; k2++)
				if (e2[k2] != NO_NODE) cc_claim(C, (uint32_t)v, e2[k2] >> 1);
		}
	}
}

static void cc_run(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	cc_ctx *C = (cc_ctx *)vc;
	uint64_t off = 0;
	tl_dirty.on = 1;
	tl_dirty.v = NULL;
	tl_dirty.n = tl_dirty.cap = 0;
	for (uint64_t c = lo; c < hi; c++)
		for (uint64_t k = C->cstart[c]; k < C->cstart[c + 1]; k++)
			visit_minor_out(C->g, &C->g->nodes[C->ex[C->order[k]]], C->threshold, &off, NULL);
	tl_dirty.on = 0;
	const int slot = __sync_fetch_and_add(&C->ntl, 1);
	C->tl[slot] = tl_dirty.v;
	C->tln[slot] = tl_dirty.n;
	__sync_fetch_and_add(&C->off, off);
}

/* returns 0 when the pass was committed here, 1 when the caller has to do it the sequential way */
static int commit_minor_out_by_components(graph_t *g, const uint64_t *ex, uint64_t nexec, double threshold, uint64_t *off)
{
	if (nexec < 4096 || nexec > 0xFFFFFFF0ULL || par_threads() < 2) return 1;
	cc_ctx C;
	memset(&C, 0, sizeof C);
	C.g = g; C.ex = ex; C.nexec = nexec; C.threshold = threshold;
	C.owner = (uint32_t *)malloc((g->n + 1) * sizeof(uint32_t));
	C.parent = (uint32_t *)malloc((nexec + 1) * sizeof(uint32_t));
	if (!C.owner || !C.parent) { free(C.owner); free(C.parent); return 1; }
	double t_sub = cut_now_ms();
	par_for(0, g->n + 1, 1 << 20, cc_clear, C.owner);
	for (uint64_t v = 0; v < nexec; v++) C.parent[v] = (uint32_t)v;
	SUBPHASE("  components: clear");
	par_for(0, nexec, 2048, cc_link, &C);
	free(C.owner);
	SUBPHASE("  components: union");
	/* group: component id = its smallest visit (roots are minima), members in ascending order */
	uint64_t *count = (uint64_t *)calloc(nexec + 1, sizeof(uint64_t));
	uint32_t *root = (uint32_t *)malloc((nexec + 1) * sizeof(uint32_t));
	for (uint64_t v = 0; v < nexec; v++) { root[v] = cc_find(C.parent, (uint32_t)v); count[root[v]]++; }
	C.cstart = (uint64_t *)malloc((nexec + 2) * sizeof(uint64_t));
	uint64_t *where = (uint64_t *)malloc((nexec + 1) * sizeof(uint64_t));
	uint64_t pos = 0;
	for (uint64_t v = 0; v < nexec; v++)
		if (count[v]) { where[v] = pos; C.cstart[C.ncomp++] = pos; pos += count[v]; }
	C.cstart[C.ncomp] = pos;
	C.order = (uint64_t *)malloc((nexec + 1) * sizeof(uint64_t));
	for (uint64_t v = 0; v < nexec; v++) C.order[where[root[v]]++] = v;
	free(count); free(root); free(where); free(C.parent);
	SUBPHASE("  components: group");
	/* dynamic chunks of a few components (transcripts differ wildly in size); one dirty list per chunk */
	const uint64_t per = 16;
	const uint64_t nchunks = (C.ncomp + per - 1) / per;
	C.tl = (uint64_t **)calloc(nchunks + 1, sizeof(uint64_t *));
	C.tln = (size_t *)calloc(nchunks + 1, sizeof(size_t));
	uint64_t biggest = 0;
	for (uint64_t c = 0; c < C.ncomp; c++) if (C.cstart[c + 1] - C.cstart[c] > biggest) biggest = C.cstart[c + 1] - C.cstart[c];
	par_for(0, C.ncomp, per, cc_run, &C);
	SUBPHASE("  components: run");
	for (int t = 0; t < C.ntl; t++) {
		for (size_t k = 0; k < C.tln[t]; k++) {
			const uint64_t i = C.tl[t][k];
			if (g->dn == g->dcap) {
				g->dcap = g->dcap ? g->dcap * 2 : 4096;
				g->dlist = (uint64_t *)realloc(g->dlist, g->dcap * sizeof(uint64_t));
			}
			g->dlist[g->dn++] = i;
		}
		free(C.tl[t]);
	}
	if (sdt_env("SDT_TIMING")) fprintf(stderr, "[cuttip]     %llu visits in %llu components, largest %llu\n", (unsigned long long)nexec, (unsigned long long)C.ncomp, (unsigned long long)biggest);
	free(C.tl); free(C.tln); free(C.order); free(C.cstart);
	*off += C.off;
	return 0;
}

/* ---- the device's labelled records (sdt_gpu_minor_out_labelled): MO_RW words = node, 8 neighbours, their 8 occurrence counts
 * (two per word), component label; the junction
 * records come sorted by (label, node), so a component is a run of records and its visits are in the reference's order ---- */
static void mo_scatter_records(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	void **a = (void **)vc;
	graph_t *g = (graph_t *)a[0];
	const uint64_t *rec = (const uint64_t *)a[1];
	for (uint64_t r = lo; r < hi; r++) {
		memcpy(&g->nb_pool[r * 8], &rec[r * MO_RW + 1], 8 * sizeof(uint64_t));
		memcpy(&g->nb_cnt[r * 8], &rec[r * MO_RW + 9], 8 * sizeof(uint32_t));
		g->nb_slot[rec[r * MO_RW]] = (uint32_t)(r + 1);
	}
}

typedef struct {
	graph_t *g;
	const uint64_t *rec;
	const uint64_t *cstart;
	const uint32_t *corder;
	double threshold;
	volatile uint64_t off;
	uint64_t *tl[64];         /* dirty lists, one per thread */
	size_t tln[64], tlcap[64];
} ml_run_ctx;

static void mo_run_labelled(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	ml_run_ctx *C = (ml_run_ctx *)vc;
	uint64_t off = 0;
	tl_dirty.on = 1;
	tl_dirty.v = C->tl[tid];                              /* one list per thread, kept across its chunks */
	tl_dirty.n = C->tln[tid];
	tl_dirty.cap = C->tlcap[tid];
	graph_t *g = C->g;
	for (uint64_t k = lo; k < hi; k++) {
		const uint64_t c = C->corder[k];
		const uint64_t r1 = C->cstart[c + 1];
		for (uint64_t r = C->cstart[c]; r < r1; r++) {
			/* a visit is a chain of dependent cache misses (junction -> neighbours -> the cut neighbour's own neighbours); the
			 * records lie in visiting order and name every neighbour, so the lines of the visits ahead are asked for now:
			 * three visits ahead the junction and its neighbours (node, pool slot, dirty mark), one visit ahead the neighbour
			 * tables of those neighbours */
			if (r + 3 < r1) {
				const uint64_t *R = C->rec + (r + 3) * MO_RW;
				__builtin_prefetch(&g->nodes[R[0]], 1);
				for (int q = 1; q <= 8; q++)
					if (R[q] != NO_NODE) {
						__builtin_prefetch(&g->nodes[R[q] >> 1], 1);
						__builtin_prefetch(&g->nb_slot[R[q] >> 1]);
						__builtin_prefetch(&g->dirty[R[q] >> 1], 1);
					}
			}
			if (r + 1 < r1) {
				const uint64_t *R = C->rec + (r + 1) * MO_RW;
				for (int q = 1; q <= 8; q++)
					if (R[q] != NO_NODE) {
						const uint32_t sl = g->nb_slot[R[q] >> 1];
						if (sl) __builtin_prefetch(&g->nb_pool[(uint64_t)(sl - 1) * 8]);
					}
			}
			visit_minor_out(g, &g->nodes[C->rec[r * MO_RW]], C->threshold, &off, NULL);
		}
	}
	tl_dirty.on = 0;
	C->tl[tid] = tl_dirty.v;
	C->tln[tid] = tl_dirty.n;
	C->tlcap[tid] = tl_dirty.cap;
	__sync_fetch_and_add(&C->off, off);
}

static void commit_minor_out_labelled(graph_t *g, const uint64_t *rec, uint64_t nj, double threshold, uint64_t *off)
{
	if (!nj) return;
	uint64_t ncomp = 1;
	for (uint64_t r = 1; r < nj; r++) ncomp += rec[r * MO_RW + MO_RW - 1] != rec[r * MO_RW - 1];
	if (ncomp > 0xFFFFFFF0ULL) { printf("too many components of junctions\n"); exit(1); }
	uint64_t *cstart = (uint64_t *)malloc((ncomp + 1) * sizeof(uint64_t));
	ncomp = 0;
	cstart[0] = 0;
	for (uint64_t r = 1; r < nj; r++)
		if (rec[r * MO_RW + MO_RW - 1] != rec[r * MO_RW - 1]) cstart[++ncomp] = r;
	cstart[++ncomp] = nj;
	uint32_t *corder = (uint32_t *)malloc((ncomp + 1) * sizeof(uint32_t));          /* largest first */
	uint64_t bucket[66] = {0}, biggest = 0;
	for (uint64_t c = 0; c < ncomp; c++) {
		const uint64_t sz = cstart[c + 1] - cstart[c];
		bucket[64 - __builtin_clzll(sz)]++;
		if (sz > biggest) biggest = sz;
	}
	{
		uint64_t acc = 0;
		for (int b = 65; b >= 0; b--) { const uint64_t t = bucket[b]; bucket[b] = acc; acc += t; }
	}
	for (uint64_t c = 0; c < ncomp; c++) corder[bucket[64 - __builtin_clzll(cstart[c + 1] - cstart[c])]++] = (uint32_t)c;
	ml_run_ctx C;
	memset(&C, 0, sizeof C);
	C.g = g; C.rec = rec; C.cstart = cstart; C.corder = corder; C.threshold = threshold;
	/* largest first: a few thousand components (the long ones the device left to the host) go out one by one -- sixteen of the
	 * largest in one task were 300 K visits on one thread, 0.42 s of a 0.3 s job */
	par_for(0, ncomp, ncomp < 65536 ? 1 : 16, mo_run_labelled, &C);
	for (int t = 0; t < 64; t++) {
		for (size_t k = 0; k < C.tln[t]; k++) {
			if (g->dn == g->dcap) {
				g->dcap = g->dcap ? g->dcap * 2 : 4096;
				g->dlist = (uint64_t *)realloc(g->dlist, g->dcap * sizeof(uint64_t));
			}
			g->dlist[g->dn++] = C.tl[t][k];
		}
		free(C.tl[t]);
	}
	if (sdt_env("SDT_TIMING")) fprintf(stderr, "[cuttip]     %llu visits in %llu components, largest %llu\n", (unsigned long long)nj, (unsigned long long)ncomp, (unsigned long long)biggest);
	free(cstart); free(corder);
	*off += C.off;
}

typedef struct { graph_t *g; const uint64_t *node; const uint32_t *l, *r; } aw_ctx;

static void apply_written_part(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	aw_ctx *A = (aw_ctx *)vc;
	for (uint64_t k = lo; k < hi; k++) {
		if (k + 16 < hi) __builtin_prefetch(&A->g->nodes[A->node[k + 16]], 1);
		gnode_t *n = &A->g->nodes[A->node[k]];
		n->l_links = A->l[k] & 0xFFFFFFu;
		n->r_links = A->r[k] & 0xFFFFFFu;
		n->linear = (A->r[k] >> 24) & 1u;
		n->deleted = (A->r[k] >> 25) & 1u;
	}
}

void graph_apply_written(graph_t *g, const uint64_t *node, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n)
{
	aw_ctx A = {g, node, l_links, r_flags};
	par_for(0, n, 1 << 14, apply_written_part, &A);
}

uint64_t graph_remove_minor_out(graph_t *g, int dd)
{
	const double threshold = (double)dd / 100;
	uint64_t off = 0;
	printf("Start to remove kmer of out frequency kmers < %f\n", threshold);
	double t_sub = cut_now_ms();
	if (g->dev_minor_out_commit_begin && !sdt_test_env("SDT_HOST_COMMIT")) {
		/* dry run, components, commit and re-marking on the device mirror; the long components here, at the same time */
		uint64_t lin = 0, *sk = NULL, nsk = 0, nskr = 0;
		if (g->dev_minor_out_commit_begin(g, threshold, &sk, &nsk, &nskr) != 0) {
			printf("the device commit failed. Now exit to system...\n");          /* no silent host fallback */
			exit(1);
		}
		graph_clear_dirty(g);          /* the hook brought the mirror up to date */
		g->dn = 0;
		SUBPHASE("device dry run + components");
		if (nsk) {
			/* the components the device leaves alone (one lane is no match for a thread on thousands of dependent visits): the
			 * labelled commit on their records (junctions first, then the neighbours they may cut).  What this writes is on the
			 * dirty list: the mirror hears about it before the next dry run */
			if (nskr > 0xFFFFFFF0ULL) { printf("too many junction records\n"); exit(1); }
			g->nb_slot = (uint32_t *)calloc(g->n + 1, sizeof(uint32_t));
			g->nb_pool = (uint64_t *)malloc((nskr + 1) * 8 * sizeof(uint64_t));
			g->nb_cnt = (uint32_t *)malloc((nskr + 1) * 8 * sizeof(uint32_t));
			void *sa[2] = {g, sk};
			par_for(0, nskr, 4096, mo_scatter_records, sa);
			SUBPHASE("long components: neighbour tables");
			commit_minor_out_labelled(g, sk, nsk, threshold, &off);
			SUBPHASE("long components: commit");
			graph_free_later(g->nb_slot, g->nb_pool, sk, g->nb_cnt);
			g->nb_slot = NULL;
			g->nb_pool = NULL;
			g->nb_cnt = NULL;
			ml_ctx M;
			memset(&M, 0, sizeof M);
			M.g = g;
			par_for(0, g->dn, 1 << 12, mark_linear_dirty, &M);
			for (int t = 0; t < 64; t++) lin += M.n[t];
			SUBPHASE("long components: free + mark linear");
		}
		uint64_t off_dev = 0, lin_dev = 0;
		if (g->dev_minor_out_commit_finish(g, &off_dev, &lin_dev) != 0) {
			printf("the device commit failed. Now exit to system...\n");
			exit(1);
		}
		off += off_dev;
		lin += lin_dev;
		SUBPHASE("device commit finished + applied");
		printf("%llu kmers off\n", (unsigned long long)off);
		printf("%d thread created for cutTipPreGraph\n", g->p);
		printf("%llu linear nodes\n", (unsigned long long)lin);
		return off;
	}
	if (g->dev_minor_out) {
		/* junction dry run + neighbour look-ups + components answered by the device mirror of the graph (sdt_gpu_minor_out_dry).
		 * Only the junctions the dry run flagged are visited: links only disappear during the pass, so on any junction the live
		 * largest count per side is <= the dry run's and every live ratio >= the dry run's -- a junction with nothing under the
		 * threshold then has nothing under it now, whatever was written around it. */
		uint64_t *rec = NULL, nj = 0, nr = 0;
		if (g->dev_minor_out(g, threshold, &rec, &nj, &nr) != 0) {
			printf("the device dry run failed. Now exit to system...\n");       /* no silent host fallback */
			exit(1);
		}
		graph_clear_dirty(g);          /* the hook brought the mirror up to date */
		g->dn = 0;
		SUBPHASE("device dry run");
		if (nr > 0xFFFFFFF0ULL) { printf("too many junction records\n"); exit(1); }
		g->nb_slot = (uint32_t *)calloc(g->n + 1, sizeof(uint32_t));
		g->nb_pool = (uint64_t *)malloc((nr + 1) * 8 * sizeof(uint64_t));
		g->nb_cnt = (uint32_t *)malloc((nr + 1) * 8 * sizeof(uint32_t));
		void *sa[2] = {g, rec};
		par_for(0, nr, 4096, mo_scatter_records, sa);
		SUBPHASE("scatter records");
		commit_minor_out_labelled(g, rec, nj, threshold, &off);
		SUBPHASE("ordered commit");
		graph_free_later(g->nb_slot, g->nb_pool, rec, g->nb_cnt);
		g->nb_slot = NULL;
		g->nb_pool = NULL;
		g->nb_cnt = NULL;
		printf("%llu kmers off\n", (unsigned long long)off);
		mark_linear(g);
		SUBPHASE("free + mark linear");
		return off;
	}
	mo_ctx c = {g, threshold, (uint8_t *)calloc(g->n + 1, 1), (uint8_t *)calloc(g->n + 1, 1), 0, 0};
	{
	par_for(0, g->n, 16384, count_junctions, &c);
	SUBPHASE("count junctions");
	const uint64_t njunc = c.cursor;
	/* every junction has at most 8 neighbours that could be flagged; in practice far fewer are */
	uint64_t cap = njunc * 3 + 1024;
	if (cap > 0xFFFFFFF0ULL) cap = 0xFFFFFFF0ULL;
	g->nb_slot = (uint32_t *)calloc(g->n + 1, sizeof(uint32_t));
	g->nb_pool = (uint64_t *)malloc(cap * 8 * sizeof(uint64_t));
	c.cursor = 0;
	c.cap = (uint32_t)cap;
	SUBPHASE("alloc");
	par_for(0, g->n, 8192, mo_junctions, &c);
	SUBPHASE("junction dry run");
	par_for(0, g->n, 16384, mo_candidates, &c);
	SUBPHASE("candidate neighbours");
	}
	/* the pass itself, in the reference's order.  Only the junctions the dry run flagged are visited: links only
	 * disappear during the pass, so on any junction the live largest count per side is <= the dry run's and every
	 * live ratio >= the dry run's -- a junction with nothing under the threshold then has nothing under it now,
	 * whatever was written around it.  neighbour() finds the answers of the executed visits precomputed. */
	/* The commit is a chain of dependent cache misses per visit (node -> its neighbour slots -> the cut
	 * neighbours -> their neighbour slots -> the nodes those unlink).  The visits the dry run flagged are known up
	 * front, so run a software prefetch pipeline a few visits ahead of the sweep. */
	uint64_t nexec = 0;
	for (uint64_t i = 0; i < g->n; i++) nexec += c.writes[i];
	uint64_t *ex = (uint64_t *)malloc((nexec + 1) * sizeof(uint64_t));
	nexec = 0;
	for (uint64_t i = 0; i < g->n; i++) if (c.writes[i]) ex[nexec++] = i;
	/* every visit and every node it may cut has its neighbour table (no pool overflow)? then by components */
	const int complete = c.cursor <= c.cap;
	const int sequential = !complete || sdt_test_env("SDT_SEQUENTIAL_COMMIT") || commit_minor_out_by_components(g, ex, nexec, threshold, &off);
	for (uint64_t cur = 0; sequential && cur < nexec; cur++) {
		/* stage 1 (far): slot number; stage 2: the 8 neighbour entries; stage 3: the neighbour nodes and their
		 * slot numbers; stage 4 (near): the neighbours' own neighbour entries */
		if (cur + 24 < nexec) __builtin_prefetch(&g->nb_slot[ex[cur + 24]]);
		if (cur + 16 < nexec) { const uint32_t sl = g->nb_slot[ex[cur + 16]]; if (sl) __builtin_prefetch(&g->nb_pool[(uint64_t)(sl - 1) * 8]); __builtin_prefetch(&g->nodes[ex[cur + 16]]); }
		if (cur + 8 < nexec) {
			const uint32_t sl = g->nb_slot[ex[cur + 8]];
			if (sl) for (int k = 0; k < 8; k++) { const uint64_t v = g->nb_pool[(uint64_t)(sl - 1) * 8 + k]; if (v != NO_NODE) { __builtin_prefetch(&g->nodes[v >> 1]); __builtin_prefetch(&g->nb_slot[v >> 1]); } }
		}
		if (cur + 4 < nexec) {
			const uint32_t sl = g->nb_slot[ex[cur + 4]];
			if (sl) for (int k = 0; k < 8; k++) { const uint64_t v = g->nb_pool[(uint64_t)(sl - 1) * 8 + k]; if (v != NO_NODE) { const uint32_t s2 = g->nb_slot[v >> 1]; if (s2) __builtin_prefetch(&g->nb_pool[(uint64_t)(s2 - 1) * 8]); } }
		}
		visit_minor_out(g, &g->nodes[ex[cur]], threshold, &off, NULL);
	}
	free(ex);
	g->touched = NULL;
	g->tn = 0;
	SUBPHASE("ordered commit");
	graph_free_later(g->nb_slot, g->nb_pool, c.need, c.writes);
	g->nb_slot = NULL;
	g->nb_pool = NULL;
	printf("%llu kmers off\n", (unsigned long long)off);
	mark_linear(g);
	SUBPHASE("free + mark linear");
	return off;
}

/* what a walk from a dead end found: everything here depends only on nodes that cannot change during a sweep
 * (the chain), so a dry run may compute it once and the ordered commit can reuse it */
typedef struct {
	uint64_t end;          /* index of the node the walk stopped at, NO_NODE = no decision to take */
	uint8_t ch, sm;        /* base by which that node sees the chain; strand on which it was reached */
	uint8_t thin_stop;     /* THIN: stopped at a linear node that is not single (:163-166) */
} walk_t;

/* the walk of clipTipFromNode (:43-281).  Returns 0 when there is nothing to decide: not a dead end, or the
 * chain is longer than cut_len. */
static int walk_from_tip(graph_t *g, const gnode_t *tip, int cut_len, int thin, walk_t *w)
{
	const int K = g->K;
	w->end = NO_NODE;
	if (tip->linear || tip->deleted) return 0;
	if (thin && !tip->single) return 0;
	const int in = degree(tip, LEFT), out = degree(tip, RIGHT);
	kw_t at;                      /* oriented k-mer we are standing on; the walk always moves "forward" from it */
	unsigned b;
	if (in == 0 && out == 1) {
		at = tip->seq;
		for (b = 0; b < 4 && !link_of(tip, RIGHT, b); b++) ;
	} else if (in == 1 && out == 0) {
		at = kw_rc(tip->seq, K);
		for (b = 0; b < 4 && !link_of(tip, LEFT, b); b++) ;
		b ^= 2u;
	} else {
		return 0;
	}
	int steps = 1, sm;
	kw_t step = kw_next(at, b, K);
	gnode_t *o = graph_find_oriented(g, step, &sm);
	w->thin_stop = 0;
	while (o->linear) {
		steps++;
		if (thin && !o->single) { w->thin_stop = 1; break; }
		if (steps > cut_len) return 0;
		at = step;                                                       /* oriented word of o */
		if (sm) { for (b = 0; b < 4 && !link_of(o, RIGHT, b); b++) ; }
		else { for (b = 0; b < 4 && !link_of(o, LEFT, b); b++) ; b ^= 2u; }
		step = kw_next(at, b, K);
		o = graph_find_oriented(g, step, &sm);
	}
	w->end = (uint64_t)(o - g->nodes);
	w->ch = (uint8_t)kw_first(&at, K);                                   /* base by which o sees the chain */
	w->sm = (uint8_t)sm;
	return 1;
}

/* the decision at the end of the walk (:282-336); reads and writes only `tip` and the end node */
static int decide_tip(graph_t *g, gnode_t *tip, const walk_t *w, int thin, uint64_t *tips, int dry)
{
	gnode_t *o = &g->nodes[w->end];
	const unsigned ch = w->ch;
	const int sm = w->sm;
	if (dry) {                                                          /* would this decision write? (no side effects) */
		if (degree(o, LEFT) + degree(o, RIGHT) == 1 || thin) return 1;
		const int side = sm ? LEFT : RIGHT;
		unsigned strongest = 0;
		for (unsigned c = 0; c < 4; c++)
			if (link_of(o, side, c) > strongest) strongest = link_of(o, side, c);
		return link_of(o, side, sm ? ch : ch ^ 2u) < strongest;
	}
	if (degree(o, LEFT) + degree(o, RIGHT) == 1) {                       /* the whole path is an island */
		(*tips)++;
		tip->deleted = 1;
		o->deleted = 1;
		touch(g, tip); touch(g, o);
		return 1;
	}
	if (thin) {
		(*tips)++;
		tip->deleted = 1;
		unlink_prev(o, ch, sm);
		o->linear = 0;
		touch(g, tip); touch(g, o);
		return 1;
	}
	const int side = sm ? LEFT : RIGHT;                                  /* side of o the chain enters */
	unsigned strongest = 0;
	for (unsigned c = 0; c < 4; c++)
		if (link_of(o, side, c) > strongest) strongest = link_of(o, side, c);
	if (link_of(o, side, sm ? ch : ch ^ 2u) < strongest) {
		(*tips)++;
		tip->deleted = 1;
		unlink_prev(o, ch, sm);
		if (one_in_one_out(o)) o->linear = 1;
		touch(g, tip); touch(g, o);
		return 1;
	}
	return 0;
}

/* one dead end, live: returns 1 when something was cut */
static int clip_tip(graph_t *g, gnode_t *tip, int cut_len, int thin, uint64_t *tips)
{
	walk_t w;
	if (!walk_from_tip(g, tip, cut_len, thin, &w)) return 0;
	return decide_tip(g, tip, &w, thin, tips, 0);
}

typedef struct { graph_t *g; int cut_len, thin; walk_t *walks; volatile uint64_t would_cut; } tips_ctx;

static void spec_tips(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tips_ctx *c = (tips_ctx *)vc;
	uint64_t cuts = 0;
	for (uint64_t i = lo; i < hi; i++)
		if (walk_from_tip(c->g, &c->g->nodes[i], c->cut_len, c->thin, &c->walks[i]))
			cuts += (uint64_t)decide_tip(c->g, &c->g->nodes[i], &c->walks[i], c->thin, NULL, 1);
	if (cuts) __sync_fetch_and_add(&c->would_cut, cuts);
}

/* the ordered part of a sweep: `stale[i]` = node i has been written since `walks` were taken */
static int commit_tips(graph_t *g, uint64_t lo, uint64_t hi, int cut_len, int thin, uint64_t *tips, const walk_t *walks,
                       const uint8_t *stale)
{
	int clipped = 0;
	for (uint64_t i = lo; i < hi; i++) {
		gnode_t *tip = &g->nodes[i];
		const walk_t *w = &walks[i];
		if (i + 64 < hi && walks[i + 64].end != NO_NODE)
			__builtin_prefetch(&g->nodes[walks[i + 64].end]);                /* the decision reads the end node */
		if (stale[i]) {
			clipped += clip_tip(g, tip, cut_len, thin, tips);
		} else if (w->end != NO_NODE) {
			if (!w->thin_stop && g->nodes[w->end].linear)
				clipped += clip_tip(g, tip, cut_len, thin, tips);
			else
				clipped += decide_tip(g, tip, w, thin, tips, 0);
		}
	}
	return clipped;
}

/* One sweep over nodes [lo, hi) with the reference's semantics; returns the number of clips.
 * Dry run: every walk, in parallel, on the graph as the sweep finds it.  Commit, in order:
 *   - a node written since the dry run is visited for real (it may have become a dead end, or stopped being one);
 *   - a walk whose end node was non-linear and has been made linear since would now run on: visited for real;
 *   - otherwise the recorded walk is still what the reference would walk (chains cannot change inside a
 *     sweep), and only the O(1) decision at its end node is taken, on the live state of that node. */
static int sweep_tips(graph_t *g, uint64_t lo, uint64_t hi, int cut_len, int thin, uint64_t *tips, tips_ctx *c, uint8_t *marks)
{
	c->cut_len = cut_len;
	c->thin = thin;
	g->touched = NULL;
	c->would_cut = 0;
	par_for(lo, hi, 8192, spec_tips, c);
	if (c->would_cut == 0)
		return 0;                                        /* nothing writes on the untouched graph => the sweep is a no-op */
	g->touched = marks;                                  /* all zero on entry (calloc / un-marked below) */
	g->tn = 0;
	const int clipped = commit_tips(g, lo, hi, cut_len, thin, tips, c->walks, marks);
	for (size_t k = 0; k < g->tn; k++) marks[g->tlist[k]] = 0;
	g->tn = 0;
	g->touched = NULL;
	return clipped;
}

/* ---- tip passes on the walks of the device: commit by components ------------------------------------------------
 * A visit of a dead end reads and writes its tip and the node its walk ends at; a cut there may turn the end node
 * into a dead end of its own (it then walks on to ITS neighbours along chains of at most cut_len nodes) or make it
 * linear (walks that stopped at it now run on to the next junction).  Whatever a pass does therefore stays inside
 * one connected component of the graph whose vertices are the non-linear nodes and whose edges are the chains of
 * linear nodes no longer than cut_len between them (removeSingleTips: the walks themselves are enough -- a single
 * k-mer has at most one link per side, so an end node that is written never starts a THIN walk that was not recorded).
 * The dry run labels every recorded walk with its component (union-find on the device) and returns the records sorted
 * by (label, node): a component is a run of records, components run side by side, each one in the reference's order
 * -- "set 0 until nothing changes, then set 1, ..." restricted to the component is exactly what the full sweeps do to
 * it, because a sweep that clips nothing in a component leaves it untouched.
 *
 * Inside a component the sweep visits, ascending: the recorded walks (static), plus every node written during the pass
 * (touch(): the first write makes it a member) -- one written AHEAD of the sweep position is still visited in this
 * sweep (min-heap), one written behind it waits for the next sweep of its set, one in an earlier set is never
 * visited again (the reference does not return to a finished set). */
#define CW_NODE(rec, r) ((rec)[3 * (r)] & 0x00FFFFFFFFFFFFFFULL)
typedef struct { uint64_t *v; size_t n, cap; } u64vec;
static inline void vec_push(u64vec *a, uint64_t x)
{
	if (a->n == a->cap) {
		a->cap = a->cap ? a->cap * 2 : 64;
		a->v = (uint64_t *)realloc(a->v, a->cap * sizeof(uint64_t));
	}
	a->v[a->n++] = x;
}

typedef struct tip_comp {
	graph_t *g;
	const uint64_t *rec;       /* this component's walks: 3 words each, ascending node index */
	uint64_t n;
	uint64_t pos, lo, hi;      /* node being visited; range of the current sweep */
	int sweeping;
	u64vec heap;               /* written ahead of pos (this set or a later one): min-heap */
	u64vec behind;             /* written behind pos inside [lo, hi) */
	u64vec dyn;                /* written nodes of the current set that have been visited once, ascending */
	u64vec vis;                /* popped from the heap during this sweep, ascending */
} tip_comp;

static void heap_push(u64vec *h, uint64_t x)
{
	vec_push(h, x);
	size_t i = h->n - 1;
	while (i) {
		const size_t p = (i - 1) >> 1;
		if (h->v[p] <= h->v[i]) break;
		const uint64_t t = h->v[p]; h->v[p] = h->v[i]; h->v[i] = t;
		i = p;
	}
}
static uint64_t heap_pop(u64vec *h)
{
	const uint64_t top = h->v[0];
	h->v[0] = h->v[--h->n];
	size_t i = 0;
	for (;;) {
		size_t l = 2 * i + 1, r = l + 1, m = i;
		if (l < h->n && h->v[l] < h->v[m]) m = l;
		if (r < h->n && h->v[r] < h->v[m]) m = r;
		if (m == i) break;
		const uint64_t t = h->v[m]; h->v[m] = h->v[i]; h->v[i] = t;
		i = m;
	}
	return top;
}

static int comp_is_static(const tip_comp *C, uint64_t node)
{
	uint64_t lo = 0, hi = C->n;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (CW_NODE(C->rec, mid) < node) lo = mid + 1; else hi = mid;
	}
	return lo < C->n && CW_NODE(C->rec, lo) == node;
}

/* touch() calls this the first time a node is written in the pass */
static void comp_first_touch(tip_comp *C, uint64_t i)
{
	if (comp_is_static(C, i)) return;                     /* a recorded walk: visited anyway, live once it is dirty */
	if (i > C->pos) heap_push(&C->heap, i);
	else if (i < C->pos && i >= C->lo && C->sweeping) vec_push(&C->behind, i);
}

static int cmp_u64(const void *a, const void *b)
{
	const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return x < y ? -1 : x > y;
}

/* one sweep of [lo, hi) over the component; kc = first static record at or after lo */
static int comp_sweep(tip_comp *C, uint64_t kc, int cut_len, int thin, uint64_t *tips)
{
	graph_t *g = C->g;
	int clipped = 0;
	size_t kd = 0;
	C->sweeping = 1;
	for (;;) {
		const uint64_t a = kc < C->n && CW_NODE(C->rec, kc) < C->hi ? CW_NODE(C->rec, kc) : NO_NODE;
		const uint64_t b = kd < C->dyn.n ? C->dyn.v[kd] : NO_NODE;
		const uint64_t h = C->heap.n && C->heap.v[0] < C->hi ? C->heap.v[0] : NO_NODE;
		uint64_t m = a < b ? a : b;
		if (h < m) m = h;
		if (m == NO_NODE) break;
		C->pos = m;
		gnode_t *tip = &g->nodes[m];
		if (m == h) {
			heap_pop(&C->heap);
			vec_push(&C->vis, m);
			clipped += clip_tip(g, tip, cut_len, thin, tips);
		} else if (m == b) {
			kd++;
			clipped += clip_tip(g, tip, cut_len, thin, tips);
		} else {
			if (kc + 4 < C->n) __builtin_prefetch(&g->nodes[C->rec[3 * (kc + 4) + 1]]);       /* the decision reads the end node */
			if (g->dirty[m]) {
				clipped += clip_tip(g, tip, cut_len, thin, tips);
			} else {
				const unsigned inf = (unsigned)(C->rec[3 * kc] >> 56);
				walk_t wk;
				wk.end = C->rec[3 * kc + 1];
				wk.ch = (uint8_t)(inf & 3u);
				wk.sm = (uint8_t)((inf >> 2) & 1u);
				wk.thin_stop = (uint8_t)((inf >> 3) & 1u);
				if (!wk.thin_stop && g->nodes[wk.end].linear)
					clipped += clip_tip(g, tip, cut_len, thin, tips);            /* the end node has become linear: the walk runs on */
				else
					clipped += decide_tip(g, tip, &wk, thin, tips, 0);
			}
			kc++;
		}
	}
	C->sweeping = 0;
	/* the written nodes of this range, for the next sweep of it: dyn + vis (both ascending) + behind */
	if (C->vis.n || C->behind.n) {
		for (size_t k = 0; k < C->vis.n; k++) vec_push(&C->dyn, C->vis.v[k]);
		for (size_t k = 0; k < C->behind.n; k++) vec_push(&C->dyn, C->behind.v[k]);
		qsort(C->dyn.v, C->dyn.n, sizeof(uint64_t), cmp_u64);
		C->vis.n = C->behind.n = 0;
	}
	return clipped;
}

typedef struct {
	graph_t *g;
	const uint64_t *rec;
	const uint64_t *cstart;
	const uint32_t *corder;
	uint64_t ncomp;
	int thin, cut_len;
	volatile uint64_t tips;
	uint64_t *tl[64];         /* dirty lists, one per thread */
	size_t tln[64], tlcap[64];
} tc_ctx;

static void tc_run(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	tc_ctx *T = (tc_ctx *)vc;
	graph_t *g = T->g;
	uint64_t tips = 0;
	tip_comp C;
	memset(&C, 0, sizeof C);
	C.g = g;
	tl_dirty.on = 1;
	tl_dirty.v = T->tl[tid];                              /* one list per thread, kept across its chunks */
	tl_dirty.n = T->tln[tid];
	tl_dirty.cap = T->tlcap[tid];
	tl_comp = &C;
	for (uint64_t k = lo; k < hi; k++) {
		const uint64_t c = T->corder[k];
		/* almost all components hold one to four walks, so the prefetch inside comp_sweep (four records ahead) never fires: the
		 * nodes of the components eight further on are asked for here (components of a size class come in ascending order:
		 * cstart[] and rec[] are read almost sequentially, nodes[] and dirty[] are the random accesses) */
		if (k + 8 < hi) {
			const uint64_t c8 = T->corder[k + 8];
			const uint64_t *r8 = T->rec + 3 * T->cstart[c8];
			uint64_t n8 = T->cstart[c8 + 1] - T->cstart[c8];
			if (n8 > 4) n8 = 4;
			for (uint64_t j = 0; j < n8; j++) {
				const uint64_t tipn = CW_NODE(r8, j);
				__builtin_prefetch(&g->nodes[tipn]);
				__builtin_prefetch((const char *)&g->nodes[tipn] + 40);
				__builtin_prefetch(&g->nodes[r8[3 * j + 1]]);
				__builtin_prefetch(&g->dirty[tipn]);
			}
		}
		C.rec = T->rec + 3 * T->cstart[c];
		C.n = T->cstart[c + 1] - T->cstart[c];
		C.heap.n = C.behind.n = C.dyn.n = C.vis.n = 0;
		if (T->thin) {                                         /* removeSingleTips: one sweep over everything */
			C.lo = 0; C.hi = g->n; C.pos = 0;
			comp_sweep(&C, 0, T->cut_len, 1, &tips);
			continue;
		}
		uint64_t kc = 0;
		int s = 0;
		while (s < g->p) {
			/* the next set this component has anything in */
			const uint64_t a = kc < C.n ? CW_NODE(C.rec, kc) : NO_NODE;
			const uint64_t h = C.heap.n ? C.heap.v[0] : NO_NODE;
			const uint64_t m = a < h ? a : h;
			if (m == NO_NODE) break;
			while (g->set_start[s + 1] <= m) s++;
			C.lo = g->set_start[s]; C.hi = g->set_start[s + 1];
			C.dyn.n = 0;
			int changed = 1;
			while (changed) {                                  /* fixed point per set before the next set (:385-408) */
				C.pos = C.lo;
				changed = comp_sweep(&C, kc, T->cut_len, 0, &tips);
			}
			while (kc < C.n && CW_NODE(C.rec, kc) < C.hi) kc++;
			s++;
		}
	}
	tl_comp = NULL;
	tl_dirty.on = 0;
	free(C.heap.v); free(C.behind.v); free(C.dyn.v); free(C.vis.v);
	T->tl[tid] = tl_dirty.v;
	T->tln[tid] = tl_dirty.n;
	T->tlcap[tid] = tl_dirty.cap;
	__sync_fetch_and_add(&T->tips, tips);
}

/* the walks of every dead end from the device mirror of the graph, labelled and sorted by (component, node) */
static void device_walks(graph_t *g, int thin, int cut_len, uint64_t **rec, uint64_t *nrec)
{
	*rec = NULL;
	*nrec = 0;
	if (g->dev_walks(g, thin, cut_len, rec, nrec) != 0) {
		printf("the device dry run failed. Now exit to system...\n");       /* no silent host fallback */
		exit(1);
	}
	graph_clear_dirty(g);          /* the mirror is current as of now */
	g->dn = 0;
}

/* the set-up of commit_tips_by_components on all threads: tens of millions of records and components (the three serial loops over
 * them took a quarter of the commit) */
typedef struct { const uint64_t *rec; uint64_t nrec, nblocks, per; uint64_t *count; uint64_t *cstart; uint32_t *corder; uint64_t ncomp; uint64_t *hist; } tcs_ctx;
static void tcs_count(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tcs_ctx *S = (tcs_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		const uint64_t r0 = b * S->per > 1 ? b * S->per : 1, r1 = (b + 1) * S->per < S->nrec ? (b + 1) * S->per : S->nrec;
		uint64_t n = 0;
		for (uint64_t r = r0; r < r1; r++) n += S->rec[3 * r + 2] != S->rec[3 * r - 1];
		S->count[b] = n;
	}
}
static void tcs_fill(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tcs_ctx *S = (tcs_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		const uint64_t r0 = b * S->per > 1 ? b * S->per : 1, r1 = (b + 1) * S->per < S->nrec ? (b + 1) * S->per : S->nrec;
		uint64_t at = S->count[b];                                 /* (exclusive prefix by now) + 1: cstart[0] = 0 is the first component */
		for (uint64_t r = r0; r < r1; r++)
			if (S->rec[3 * r + 2] != S->rec[3 * r - 1]) S->cstart[1 + at++] = r;
	}
}
static void tcs_hist(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tcs_ctx *S = (tcs_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		const uint64_t c0 = b * S->per, c1 = (b + 1) * S->per < S->ncomp ? (b + 1) * S->per : S->ncomp;
		uint64_t *h = S->hist + b * 66;
		for (uint64_t c = c0; c < c1; c++) h[64 - __builtin_clzll(S->cstart[c + 1] - S->cstart[c])]++;
	}
}
static void tcs_place(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tcs_ctx *S = (tcs_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		const uint64_t c0 = b * S->per, c1 = (b + 1) * S->per < S->ncomp ? (b + 1) * S->per : S->ncomp;
		uint64_t *h = S->hist + b * 66;                            /* (first position of this block's components of every size class by now) */
		for (uint64_t c = c0; c < c1; c++) S->corder[h[64 - __builtin_clzll(S->cstart[c + 1] - S->cstart[c])]++] = (uint32_t)c;
	}
}
typedef struct { graph_t *g; uint64_t **tl; size_t *tln; size_t *at; } tcd_ctx;
static void tcd_copy(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	tcd_ctx *D = (tcd_ctx *)vc;
	for (uint64_t t = lo; t < hi; t++)
		if (D->tln[t]) memcpy(D->g->dlist + D->at[t], D->tl[t], D->tln[t] * sizeof(uint64_t));
}

static uint64_t commit_tips_by_components(graph_t *g, const uint64_t *rec, uint64_t nrec, int thin, int cut_len)
{
	if (!nrec) return 0;
	/* components = runs of equal labels */
	tcs_ctx S;
	memset(&S, 0, sizeof S);
	S.rec = rec; S.nrec = nrec;
	S.nblocks = nrec / 65536 + 1; S.per = (nrec + S.nblocks - 1) / S.nblocks;
	S.count = (uint64_t *)calloc(S.nblocks + 1, sizeof(uint64_t));
	par_for(0, S.nblocks, 1, tcs_count, &S);
	uint64_t ncomp = 1;
	for (uint64_t b = 0; b < S.nblocks; b++) { const uint64_t t = S.count[b]; S.count[b] = ncomp - 1; ncomp += t; }
	if (ncomp > 0xFFFFFFF0ULL) { printf("too many components of tips\n"); exit(1); }
	uint64_t *cstart = (uint64_t *)malloc((ncomp + 1) * sizeof(uint64_t));
	cstart[0] = 0;
	S.cstart = cstart;
	par_for(0, S.nblocks, 1, tcs_fill, &S);
	cstart[ncomp] = nrec;
	free(S.count);
	/* largest first (counting sort by the bit length of the size): the giant components must not start last */
	uint32_t *corder = (uint32_t *)malloc((ncomp + 1) * sizeof(uint32_t));
	S.ncomp = ncomp; S.corder = corder;
	S.nblocks = ncomp / 65536 + 1; S.per = (ncomp + S.nblocks - 1) / S.nblocks;
	S.hist = (uint64_t *)calloc(S.nblocks * 66 + 66, sizeof(uint64_t));
	par_for(0, S.nblocks, 1, tcs_hist, &S);
	{
		uint64_t acc = 0;
		for (int bit = 65; bit >= 0; bit--)
			for (uint64_t b = 0; b < S.nblocks; b++) { const uint64_t t = S.hist[b * 66 + bit]; S.hist[b * 66 + bit] = acc; acc += t; }
	}
	par_for(0, S.nblocks, 1, tcs_place, &S);
	free(S.hist);
	tc_ctx T;
	memset(&T, 0, sizeof T);
	T.g = g; T.rec = rec; T.cstart = cstart; T.corder = corder; T.ncomp = ncomp; T.thin = thin; T.cut_len = cut_len;
	par_for(0, ncomp, 256, tc_run, &T);
	{
		/* the threads' lists of written nodes behind what the list holds already */
		size_t at[64], total = g->dn;
		for (int t = 0; t < 64; t++) { at[t] = total; total += T.tln[t]; }
		if (total > g->dcap) {
			g->dcap = total + total / 4 + 4096;
			g->dlist = (uint64_t *)realloc(g->dlist, g->dcap * sizeof(uint64_t));
		}
		tcd_ctx D = {g, T.tl, T.tln, at};
		par_for(0, 64, 1, tcd_copy, &D);
		g->dn = total;
		for (int t = 0; t < 64; t++) free(T.tl[t]);
	}
	if (sdt_env("SDT_TIMING")) {
		uint64_t biggest = 0;
		for (uint64_t c = 0; c < ncomp; c++) if (cstart[c + 1] - cstart[c] > biggest) biggest = cstart[c + 1] - cstart[c];
		fprintf(stderr, "[cuttip]     %llu walks in %llu components, largest %llu\n", (unsigned long long)nrec, (unsigned long long)ncomp, (unsigned long long)biggest);
	}
	free(cstart); free(corder);
	return T.tips;
}

uint64_t graph_remove_single_tips(graph_t *g)
{
	uint64_t tips = 0;
	printf("Start to remove tips of single frequency kmers short than %d\n", 2 * g->K);
	double t_sub = cut_now_ms();
	if (g->dev_walks) {
		uint64_t *rec, nrec;
		device_walks(g, 1, 2 * g->K, &rec, &nrec);
		SUBPHASE("single tips: device walks");
		tips = commit_tips_by_components(g, rec, nrec, 1, 2 * g->K);
		graph_free_later(rec, NULL, NULL, NULL);           /* (a gigabyte of records: its pages go back beside the next pass) */
		SUBPHASE("single tips: commit");
	} else {
		tips_ctx c = {g, 0, 0, (walk_t *)malloc((g->n + 1) * sizeof(walk_t)), 0};
		uint8_t *marks = (uint8_t *)calloc(g->n + 1, 1);
		sweep_tips(g, 0, g->n, 2 * g->K, 1, &tips, &c, marks);
		free(marks);
		free(c.walks);
	}
	printf("%llu tips off\n", (unsigned long long)tips);
	mark_linear(g);
	return tips;
}

uint64_t graph_remove_minor_tips(graph_t *g)
{
	uint64_t tips = 0;
	printf("Start to remove tips which don't contribute the most links\n");
	double t_sub = cut_now_ms();
	if (g->dev_walks) {
		uint64_t *rec, nrec;
		device_walks(g, 0, 2 * g->K, &rec, &nrec);
		SUBPHASE("minor tips: device walks");
		tips = commit_tips_by_components(g, rec, nrec, 0, 2 * g->K);
		graph_free_later(rec, NULL, NULL, NULL);
		for (int s = 0; s < g->p; s++) printf("kmer set %d done\n", s);
	} else {
		tips_ctx c = {g, 0, 0, (walk_t *)malloc((g->n + 1) * sizeof(walk_t)), 0};
		uint8_t *marks = (uint8_t *)calloc(g->n + 1, 1);
		for (int s = 0; s < g->p; s++) {
			int changed = 1;
			while (changed)                                /* fixed point PER SET before the next set (:385-408) */
				changed = sweep_tips(g, g->set_start[s], g->set_start[s + 1], 2 * g->K, 0, &tips, &c, marks);
			printf("kmer set %d done\n", s);
		}
		free(marks);
		free(c.walks);
	}
	SUBPHASE("minor tips: sweeps");
	printf("%llu tips off\n", (unsigned long long)tips);
	mark_linear(g);
	SUBPHASE("minor tips: mark linear");
	return tips;
}

/* blocks of nodes are counted, then formatted, in parallel; a block needs the number of vertices before it for the
 * newline after every 8th */
typedef struct { graph_t *g; uint64_t block, nblocks, first; uint64_t *before; char **txt; size_t *len; } vx_ctx;

static void vx_count(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	vx_ctx *V = (vx_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		const uint64_t i0 = b * V->block, i1 = i0 + V->block < V->g->n ? i0 + V->block : V->g->n;
		uint64_t c = 0;
		for (uint64_t i = i0; i < i1; i++) c += !(V->g->nodes[i].linear || V->g->nodes[i].deleted);
		V->before[b + 1] = c;
	}
}

static inline size_t hex_u64(char *p, uint64_t v)
{
	char tmp[16];
	int n = 0;
	do { tmp[n++] = "0123456789abcdef"[v & 15]; v >>= 4; } while (v);
	for (int k = 0; k < n; k++) p[k] = tmp[n - 1 - k];
	return (size_t)n;
}

static void vx_format(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	vx_ctx *V = (vx_ctx *)vc;
	const graph_t *g = V->g;
	for (uint64_t k = lo; k < hi; k++) {
		const uint64_t b = V->first + k;
		const uint64_t i0 = b * V->block, i1 = i0 + V->block < g->n ? i0 + V->block : g->n;
		uint64_t c = V->before[b];
		const uint64_t mine = V->before[b + 1] - V->before[b];
		char *t = (char *)malloc(mine * (size_t)(17 * g->nw + 2) + 16);
		size_t o = 0;
		for (uint64_t i = i0; i < i1; i++) {
			const gnode_t *n = &g->nodes[i];
			if (n->linear || n->deleted) continue;
			c++;
			const uint64_t *w = n->seq.w;                                        /* print_kmer, kmer.c:499-516 */
			if (g->nw == 1 && !w[3]) { memcpy(t + o, "0x0 ", 4); o += 4; }
			else
				for (int q = 4 - g->nw; q < 4; q++) { o += hex_u64(t + o, w[q]); t[o++] = ' '; }
			if (c % 8 == 0) t[o++] = '\n';
		}
		V->txt[k] = t;
		V->len[k] = o;
	}
}

int graph_vertex_quiet = 0;       /* the caller prints the "vertex outputed" line itself (it writes the file ahead of its turn) */

/* output_vertex (output_pregraph.c:29-81) with print_kmer of the emulated variant (kmer.c:499-516) */
uint64_t graph_write_vertex(graph_t *g, const char *prefix)
{
	char name[4200];
	snprintf(name, sizeof name, "%s.vertex", prefix);
	FILE *fp = fopen(name, "w");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", name); exit(-1); }
	vx_ctx V;
	V.g = g;
	V.block = 1 << 18;
	V.nblocks = (g->n + V.block - 1) / V.block;
	V.before = (uint64_t *)calloc(V.nblocks + 2, sizeof(uint64_t));
	par_for(0, V.nblocks, 1, vx_count, &V);
	for (uint64_t b = 0; b < V.nblocks; b++) V.before[b + 1] += V.before[b];
	const uint64_t c = V.before[V.nblocks];
	const uint64_t wave = (uint64_t)par_threads() * 2;
	V.txt = (char **)calloc(wave, sizeof(char *));
	V.len = (size_t *)calloc(wave, sizeof(size_t));
	for (V.first = 0; V.first < V.nblocks; V.first += wave) {
		const uint64_t n = V.nblocks - V.first < wave ? V.nblocks - V.first : wave;
		par_for(0, n, 1, vx_format, &V);
		for (uint64_t k = 0; k < n; k++) { fwrite(V.txt[k], 1, V.len[k], fp); free(V.txt[k]); }
	}
	free(V.before); free(V.txt); free(V.len);
	fputc('\n', fp);
	fclose(fp);
	if (!graph_vertex_quiet) printf("%llu vertex outputed\n", (unsigned long long)c);
	return c;
}

int graph_write_basic(const char *prefix, uint64_t vertices, int K, uint64_t num_ed, int max_read_len)
{
	char name[4200];
	snprintf(name, sizeof name, "%s.preGraphBasic", prefix);
	FILE *fp = fopen(name, "w");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", name); exit(-1); }
	fprintf(fp, "VERTEX %llu K %d\n", (unsigned long long)vertices, K);
	fprintf(fp, "\nEDGEs %llu\n", (unsigned long long)num_ed);
	fprintf(fp, "\nMaxReadLen %d MinReadLen %d MaxNameLen %d\n", max_read_len, 0, 256);
	fclose(fp);
	return 0;
}

/* ---- host stand-ins for the device hooks (sdt-graphcheck, the CPU tests) ------------------------------------------
 * The commits above are driven by what sdt_gpu_minor_out_dry / sdt_gpu_tip_walks_labelled return.  Without a device the
 * same records are made here from the host's own dry runs (walk_from_tip, mo_junctions / mo_candidates) and a host
 * union-find, so that every line of the device-path commits runs -- and is compared with the reference's files -- in
 * the CPU suite.  Never used by sdt-pregraph. */
typedef struct { graph_t *g; int thin, cut_len; u64vec out[64]; uint32_t *parent; } emu_ctx;

static void emu_walk_part(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	emu_ctx *E = (emu_ctx *)vc;
	graph_t *g = E->g;
	for (uint64_t i = lo; i < hi; i++) {
		walk_t w;
		if (!walk_from_tip(g, &g->nodes[i], E->cut_len, E->thin, &w)) continue;
		vec_push(&E->out[tid], i | ((uint64_t)(w.ch | (w.sm << 2) | (w.thin_stop << 3)) << 56));
		vec_push(&E->out[tid], w.end);
		vec_push(&E->out[tid], 0);
		if (E->thin) cc_union(E->parent, (uint32_t)i, (uint32_t)w.end);
	}
}

/* the first non-linear node behind port p of n (right links 0..3 on the stored strand, left links 0..3 on the other),
 * NO_NODE when more than max_linear linear nodes lie in between */
static uint64_t emu_port_far(graph_t *g, const gnode_t *n, int p, int max_linear)
{
	const int K = g->K;
	kw_t word = p < 4 ? kw_next(n->seq, (unsigned)p, K) : kw_next(kw_rc(n->seq, K), (unsigned)(p - 4) ^ 2u, K);
	int sm, passed = 0;
	gnode_t *o = graph_find_oriented(g, word, &sm);
	while (o->linear) {
		if (++passed > max_linear) return NO_NODE;
		unsigned b;
		if (sm) { for (b = 0; b < 4 && !link_of(o, RIGHT, b); b++) ; }
		else { for (b = 0; b < 4 && !link_of(o, LEFT, b); b++) ; b ^= 2u; }
		word = kw_next(word, b, K);
		o = graph_find_oriented(g, word, &sm);
	}
	return (uint64_t)(o - g->nodes);
}

static void emu_port_union(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	emu_ctx *E = (emu_ctx *)vc;
	graph_t *g = E->g;
	for (uint64_t i = lo; i < hi; i++) {
		const gnode_t *n = &g->nodes[i];
		if (n->linear || n->deleted) continue;
		for (int p = 0; p < 8; p++) {
			if (!(p < 4 ? link_of(n, RIGHT, (unsigned)p) : link_of(n, LEFT, (unsigned)(p - 4)))) continue;
			const uint64_t far = emu_port_far(g, n, p, E->cut_len);
			if (far != NO_NODE) cc_union(E->parent, (uint32_t)i, (uint32_t)far);
		}
	}
}

static int cmp_rec3(const void *a, const void *b)
{
	const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
	if (x[2] != y[2]) return x[2] < y[2] ? -1 : 1;
	const uint64_t nx = x[0] & 0x00FFFFFFFFFFFFFFULL, ny = y[0] & 0x00FFFFFFFFFFFFFFULL;
	return nx < ny ? -1 : nx > ny;
}

static int emu_walks(graph_t *g, int thin, int cut_len, uint64_t **records, uint64_t *nr)
{
	if (g->n > 0xFFFFFFF0ULL) return 1;
	emu_ctx E;
	memset(&E, 0, sizeof E);
	E.g = g; E.thin = thin; E.cut_len = cut_len;
	E.parent = (uint32_t *)malloc((g->n + 1) * sizeof(uint32_t));
	for (uint64_t i = 0; i <= g->n; i++) E.parent[i] = (uint32_t)i;
	par_for(0, g->n, 4096, emu_walk_part, &E);
	if (!thin) par_for(0, g->n, 4096, emu_port_union, &E);
	uint64_t n = 0;
	for (int t = 0; t < 64; t++) n += E.out[t].n / 3;
	uint64_t *rec = (uint64_t *)malloc((n + 1) * 3 * sizeof(uint64_t));
	uint64_t at = 0;
	for (int t = 63; t >= 0; t--) {                           /* any order: the sort below is what the contract asks for */
		memcpy(rec + 3 * at, E.out[t].v, E.out[t].n * sizeof(uint64_t));
		at += E.out[t].n / 3;
		free(E.out[t].v);
	}
	for (uint64_t r = 0; r < n; r++) rec[3 * r + 2] = cc_find(E.parent, (uint32_t)(rec[3 * r] & 0x00FFFFFFFFFFFFFFULL));
	qsort(rec, n, 3 * sizeof(uint64_t), cmp_rec3);
	free(E.parent);
	for (size_t k = 0; k < g->dn; k++) { }                     /* (the caller clears the dirty marks) */
	*records = rec;
	*nr = n;
	return 0;
}

static int cmp_rec10(const void *a, const void *b)
{
	const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
	if (x[MO_RW - 1] != y[MO_RW - 1]) return x[MO_RW - 1] < y[MO_RW - 1] ? -1 : 1;
	return x[0] < y[0] ? -1 : x[0] > y[0];
}

static int emu_minor_out(graph_t *g, double threshold, uint64_t **records, uint64_t *nj_out, uint64_t *nr_out)
{
	if (g->n > 0xFFFFFFF0ULL) return 1;
	mo_ctx c = {g, threshold, (uint8_t *)calloc(g->n + 1, 1), (uint8_t *)calloc(g->n + 1, 1), 0, 0};
	par_for(0, g->n, 16384, count_junctions, &c);
	uint64_t cap = (uint64_t)c.cursor * 9 + 1024;
	if (cap > 0xFFFFFFF0ULL) cap = 0xFFFFFFF0ULL;
	g->nb_slot = (uint32_t *)calloc(g->n + 1, sizeof(uint32_t));
	g->nb_pool = (uint64_t *)malloc(cap * 8 * sizeof(uint64_t));
	c.cursor = 0;
	c.cap = (uint32_t)cap;
	par_for(0, g->n, 8192, mo_junctions, &c);
	par_for(0, g->n, 16384, mo_candidates, &c);
	if (c.cursor > c.cap) return 1;
	uint32_t *parent = (uint32_t *)malloc((g->n + 1) * sizeof(uint32_t));
	for (uint64_t i = 0; i <= g->n; i++) parent[i] = (uint32_t)i;
	uint64_t nj = 0, nc = 0;
	for (uint64_t i = 0; i < g->n; i++) {
		if (c.writes[i]) nj++;
		else if (g->nb_slot[i]) nc++;
	}
	uint64_t *rec = (uint64_t *)malloc((nj + nc + 1) * MO_RW * sizeof(uint64_t));
	uint64_t aj = 0, ac = nj;
	for (uint64_t i = 0; i < g->n; i++) {
		if (!g->nb_slot[i]) continue;
		uint64_t *r = &rec[(c.writes[i] ? aj++ : ac++) * MO_RW];
		r[0] = i;
		memcpy(r + 1, &g->nb_pool[(uint64_t)(g->nb_slot[i] - 1) * 8], 8 * sizeof(uint64_t));
		for (int k = 0; k < 8; k += 2) {
			const uint64_t c0 = r[1 + k] != NO_NODE ? g->nodes[r[1 + k] >> 1].count : 0, c1 = r[2 + k] != NO_NODE ? g->nodes[r[2 + k] >> 1].count : 0;
			r[9 + k / 2] = c0 | (c1 << 32);
		}
		r[MO_RW - 1] = 0;
		for (int k = 0; k < 8; k++)
			if (r[1 + k] != NO_NODE) cc_union(parent, (uint32_t)i, (uint32_t)(r[1 + k] >> 1));
	}
	for (uint64_t r = 0; r < nj + nc; r++) rec[r * MO_RW + MO_RW - 1] = cc_find(parent, (uint32_t)rec[r * MO_RW]);   /* (the neighbours' records too, as the device does) */
	qsort(rec, nj, MO_RW * sizeof(uint64_t), cmp_rec10);
	free(parent);
	free(g->nb_slot); free(g->nb_pool); free(c.need); free(c.writes);
	g->nb_slot = NULL;
	g->nb_pool = NULL;
	*records = rec;
	*nj_out = nj;
	*nr_out = nj + nc;
	return 0;
}

/* CPU stand-in for removeMinorOut's commit on the device (sdt_gpu_minor_out_commit_begin / _finish): the dry run's records made by
 * the host; the components of at most SDT_COMMIT_MAX_COMPONENT visits (default 3072, as in sdt-pregraph) are committed here and
 * now -- by the labelled commit on nodes[] itself, which is what the device's commit + graph_apply_written leave behind; their
 * dirty marks are the caller's to clear, as after a mirror sync --, the longer ones are handed to the caller: their junction records
 * in order, then the records of the neighbours they may cut.  The CPU suite runs every golden through the caller's half this way. */
static struct { uint64_t off, lin, nw, *node; uint32_t *l, *r; } emu_commit_pending;

static int emu_minor_out_commit_begin(graph_t *g, double threshold, uint64_t **skipped, uint64_t *n_skipped, uint64_t *n_skipped_records)
{
	uint64_t *rec = NULL, nj = 0, nr = 0;
	if (emu_minor_out(g, threshold, &rec, &nj, &nr) != 0) return 1;
	const uint64_t max_comp = sdt_test_env("SDT_COMMIT_MAX_COMPONENT") ? strtoull(sdt_test_env("SDT_COMMIT_MAX_COMPONENT"), NULL, 10) : 3072;
	uint8_t *big = (uint8_t *)calloc(g->n + 1, 1);                      /* by label (a node index) */
	uint64_t nbig = 0;
	for (uint64_t r0 = 0; r0 < nj;) {
		uint64_t r1 = r0 + 1;
		while (r1 < nj && rec[r1 * MO_RW + MO_RW - 1] == rec[r0 * MO_RW + MO_RW - 1]) r1++;
		if (r1 - r0 > max_comp) { big[rec[r0 * MO_RW + MO_RW - 1]] = 1; nbig += r1 - r0; }
		r0 = r1;
	}
	uint64_t nbig_all = nbig;
	for (uint64_t r = nj; r < nr; r++) nbig_all += big[rec[r * MO_RW + MO_RW - 1]];
	uint64_t *sk = (uint64_t *)malloc((nbig_all + 1) * MO_RW * sizeof(uint64_t));
	uint64_t *sm = (uint64_t *)malloc((nr - nbig_all + 1) * MO_RW * sizeof(uint64_t));
	uint64_t a = 0, b = 0, nsmall = 0;
	for (uint64_t r = 0; r < nr; r++) {                                 /* both halves keep the order: junction records first */
		const int is_big = big[rec[r * MO_RW + MO_RW - 1]];
		memcpy((is_big ? sk + a * MO_RW : sm + b * MO_RW), rec + r * MO_RW, MO_RW * sizeof(uint64_t));
		if (is_big) a++; else { b++; if (r < nj) nsmall++; }
	}
	free(rec);
	free(big);
	/* "the device": the short components, committed and re-marked -- on nodes[], whose old contents are put back afterwards: what
	 * was written goes through the same list of (index, links, flags) and graph_apply_written as the device's writes, in _finish */
	uint64_t off = 0, lin = 0;
	graph_clear_dirty(g);
	g->dn = 0;
	gnode_t *before = (gnode_t *)malloc((g->n + 1) * sizeof(gnode_t));
	memcpy(before, g->nodes, g->n * sizeof(gnode_t));
	if (b) {
		g->nb_slot = (uint32_t *)calloc(g->n + 1, sizeof(uint32_t));
		g->nb_pool = (uint64_t *)malloc((b + 1) * 8 * sizeof(uint64_t));
		g->nb_cnt = (uint32_t *)malloc((b + 1) * 8 * sizeof(uint32_t));
		void *sa[2] = {g, sm};
		par_for(0, b, 4096, mo_scatter_records, sa);
		commit_minor_out_labelled(g, sm, nsmall, threshold, &off);
		free(g->nb_slot); free(g->nb_pool); free(g->nb_cnt);
		g->nb_slot = NULL; g->nb_pool = NULL; g->nb_cnt = NULL;
		ml_ctx M;
		memset(&M, 0, sizeof M);
		M.g = g;
		par_for(0, g->dn, 1 << 12, mark_linear_dirty, &M);
		for (int t = 0; t < 64; t++) lin += M.n[t];
	}
	free(sm);
	const uint64_t nw = g->dn;
	uint64_t *wn = (uint64_t *)malloc((nw + 1) * sizeof(uint64_t));
	uint32_t *wl = (uint32_t *)malloc((nw + 1) * sizeof(uint32_t)), *wr = (uint32_t *)malloc((nw + 1) * sizeof(uint32_t));
	for (uint64_t k = 0; k < nw; k++) {
		const uint64_t i = g->dlist[k];
		const gnode_t *n = &g->nodes[i];
		wn[k] = i;
		wl[k] = n->l_links;
		wr[k] = n->r_links | ((uint32_t)n->linear << 24) | ((uint32_t)n->deleted << 25);
		g->nodes[i] = before[i];                                  /* the host's array has not heard of it yet */
	}
	free(before);
	emu_commit_pending.off = off;
	emu_commit_pending.lin = lin;
	emu_commit_pending.nw = nw;
	emu_commit_pending.node = wn; emu_commit_pending.l = wl; emu_commit_pending.r = wr;
	if (nbig) *skipped = sk; else { free(sk); *skipped = NULL; }
	*n_skipped = nbig;
	*n_skipped_records = nbig ? nbig_all : 0;
	return 0;
}

static int emu_minor_out_commit_finish(graph_t *g, uint64_t *off, uint64_t *linear)
{
	graph_apply_written(g, emu_commit_pending.node, emu_commit_pending.l, emu_commit_pending.r, emu_commit_pending.nw);
	free(emu_commit_pending.node); free(emu_commit_pending.l); free(emu_commit_pending.r);
	emu_commit_pending.node = NULL; emu_commit_pending.l = emu_commit_pending.r = NULL;
	*off = emu_commit_pending.off;
	*linear = emu_commit_pending.lin;
	return 0;
}

void graph_emulate_device_cuts(graph_t *g)
{
	if (!g->dirty) g->dirty = (uint8_t *)calloc(g->n + 1, 1);
	g->dev_walks = emu_walks;
	g->dev_minor_out = emu_minor_out;
	g->dev_minor_out_commit_begin = emu_minor_out_commit_begin;
	g->dev_minor_out_commit_finish = emu_minor_out_commit_finish;
}
