/* edges.c -- kmer2edges: from the cleaned k-mer graph to <prefix>.edge.gz (node2edge.c, output_pregraph.c:83-100).
 *
 * Visiting order = graph order (set, slot).  From every node that is neither linear nor deleted, for each base
 * with a right link (ascending), then for each base with a left link (ascending, walking the reverse strand),
 * follow `linear` nodes until the first non-linear node.  A chain of c nodes is an edge of length c-1:
 *   from = oriented k-mer of the first node, to = of the last, seq = last base of nodes 1..c-1,
 *   bal_edge = 0 when the chain reads the same on the reverse strand (its own twin), else 1 (twin = next id),
 *   cvg = 10 * (sum over interior nodes of their four LEFT link counters) / (length-1), integer division first,
 *         or 10 * first.count for length 1; capped at 16000 (MaxEdgeCov, inc/def.h:37).
 * Side effects make each edge appear once: the first node loses its link into the chain, the last node its
 * link back; interior nodes get inEdge = 1, l_links = edge id (+bal_edge when walked against their stored
 * strand) and twin; length-1 edges register their canonical (K+1)-mer in the patch table instead.
 */
#include "../../sdt_knobs.h"
#include "graph.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define MAX_EDGE_COV 16000

typedef struct { gnode_t *node; kw_t kmer; int smaller; } bead_t;
typedef struct { bead_t *b; size_t n, cap; } chain_t;

static void chain_push(chain_t *c, gnode_t *node, kw_t kmer, int smaller)
{
	if (c->n == c->cap) {
		c->cap = c->cap ? c->cap * 2 : 1024;
		c->b = (bead_t *)realloc(c->b, c->cap * sizeof(bead_t));
	}
	c->b[c->n].node = node; c->b[c->n].kmer = kmer; c->b[c->n].smaller = smaller;
	c->n++;
}

static inline unsigned rlink(const gnode_t *n, unsigned b) { return (n->r_links >> (6 * b)) & 63u; }
static inline unsigned llink(const gnode_t *n, unsigned b) { return (n->l_links >> (6 * b)) & 63u; }

/* `linear` of a node that may be the END of somebody else's chain: during the parallel stamping that thread clears one
 * of the node's links with an atomic AND on the same 32-bit word (the flag bits themselves never change there), so
 * read the word with a relaxed atomic load instead of through the bit-field */
static inline int node_is_linear(const gnode_t *n)
{
	const uint32_t w = __atomic_load_n((const uint32_t *)&n->l_links + 1, __ATOMIC_RELAXED);
	return (int)((w >> 24) & 1u);
}

static uint64_t mix_kw(const kw_t *k)
{
	uint64_t h = 0xA0761D6478BD642FULL;
	for (int i = 0; i < 4; i++) { h ^= k->w[i]; h *= 0xE7037ED1A0B428DBULL; h ^= h >> 29; }
	return h;
}

static gpatch_t *patch_slot(graph_t *g, const kw_t *k)
{
	uint64_t h = mix_kw(k) & g->patch_mask;
	while (g->patch[h].used && !kw_eq(&g->patch[h].seq, k)) h = (h + 1) & g->patch_mask;
	return &g->patch[h];
}

static void patch_put(graph_t *g, const kw_t *k, uint32_t edge, uint8_t twin)
{
	if ((g->patch_n + 1) * 2 > g->patch_mask + 1) {
		gpatch_t *old = g->patch;
		const uint64_t oldcap = g->patch_mask + 1;
		g->patch_mask = oldcap * 2 - 1;
		g->patch = (gpatch_t *)calloc(oldcap * 2, sizeof(gpatch_t));
		for (uint64_t i = 0; i < oldcap; i++)
			if (old[i].used) *patch_slot(g, &old[i].seq) = old[i];
		free(old);
	}
	gpatch_t *s = patch_slot(g, k);
	if (s->used) {
		printf("longNode %llx %llx %llx %llx already exist\n", (unsigned long long)k->w[0], (unsigned long long)k->w[1],
		       (unsigned long long)k->w[2], (unsigned long long)k->w[3]);
	} else {
		s->used = 1;
		s->seq = *k;
		g->patch_n++;
	}
	s->edge = edge;
	s->twin = twin;
}

const gpatch_t *graph_find_patch(const graph_t *g, const kw_t *k)
{
	if (!g->patch) return NULL;
	uint64_t h = mix_kw(k) & g->patch_mask;
	while (g->patch[h].used) {
		if (kw_eq(&g->patch[h].seq, k)) return &g->patch[h];
		h = (h + 1) & g->patch_mask;
	}
	return NULL;
}

static void gz_kmer(gzFile fp, const graph_t *g, const kw_t *k)
{
	const uint64_t *w = k->w;
	if (g->nw == 4) gzprintf(fp, "%llx %llx %llx %llx,", (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3]);
	else if (g->nw == 2) gzprintf(fp, "%llx %llx,", (unsigned long long)w[2], (unsigned long long)w[3]);
	else if (w[3]) gzprintf(fp, "%llx,", (unsigned long long)w[3]);
	else gzprintf(fp, "0x0,");
}

/* stringBeads (node2edge.c:58-191): extend the chain from its first bead over base `nextch` */
static void follow(graph_t *g, chain_t *c, unsigned nextch)
{
	const int K = g->K;
	int sm;
	kw_t word = kw_next(c->b[0].kmer, nextch, K);
	gnode_t *o = graph_find_oriented(g, word, &sm);
	while (node_is_linear(o)) {
		chain_push(c, o, word, sm);
		unsigned b;
		if (sm) { for (b = 0; b < 4 && !rlink(o, b); b++) ; }
		else { for (b = 0; b < 4 && !llink(o, b); b++) ; b ^= 2u; }
		word = kw_next(word, b, K);
		o = graph_find_oriented(g, word, &sm);
	}
	chain_push(c, o, word, sm);
}

/* merge_linearV2 (node2edge.c:351-561) */
static void emit_edge(graph_t *g, chain_t *c, gzFile fp, char **seqbuf, size_t *seqcap, uint64_t *plain_edges, uint64_t *extra_nodes)
{
	const int K = g->K;
	const size_t cnt = c->n;
	const int length = (int)cnt - 1;
	/* its own reverse complement?  kmer[cnt-1-i] == rc(kmer[i]) for every i (check_iden_kmerList :563-588) */
	int bal_edge = 0;
	for (size_t i = 0; i < cnt; i++) {
		kw_t r = kw_rc(c->b[i].kmer, K);
		if (!kw_eq(&c->b[cnt - 1 - i].kmer, &r)) { bal_edge = 1; break; }
	}
	bead_t *first = &c->b[0], *second = &c->b[1], *last = &c->b[cnt - 1], *second_last = &c->b[cnt - 2];
	if ((size_t)length + 1 > *seqcap) { *seqcap = (size_t)length * 2 + 16; *seqbuf = (char *)realloc(*seqbuf, *seqcap); }
	char *seq = *seqbuf;
	/* the last node forgets the chain, the first node forgets its way into it (:387-399) */
	{
		const unsigned fc = kw_first(&second_last->kmer, K);
		if (last->smaller) last->node->l_links &= ~(63u << (6 * fc));
		else last->node->r_links = last->node->r_links & ~(63u << (6 * (fc ^ 2u)));
		const unsigned lc = kw_last(&second->kmer);
		if (first->smaller) first->node->r_links = first->node->r_links & ~(63u << (6 * lc));
		else first->node->l_links &= ~(63u << (6 * (lc ^ 2u)));
	}
	long long symbol = 0;
	g->num_ed++;
	(*plain_edges)++;
	const uint32_t id = (uint32_t)g->num_ed;
	if (length == 1) {
		(*extra_nodes)++;
		/* KmerPlus(from, last base of to) = the (K+1)-mer spanning both junction k-mers; canonical over K+1 */
		kw_t plus;
		plus.w[0] = (first->kmer.w[0] << 2) | (first->kmer.w[1] >> 62);
		plus.w[1] = (first->kmer.w[1] << 2) | (first->kmer.w[2] >> 62);
		plus.w[2] = (first->kmer.w[2] << 2) | (first->kmer.w[3] >> 62);
		plus.w[3] = (first->kmer.w[3] << 2) | kw_last(&last->kmer);
		kw_t bal = kw_rc_kplus1(plus, K);
		if (kw_less(&plus, &bal)) patch_put(g, &plus, id, (uint8_t)(bal_edge + 1));
		else patch_put(g, &bal, id + (uint32_t)bal_edge, (uint8_t)(1 - bal_edge));
		symbol = first->node->count;                                      /* :474-478 */
	}
	seq[length - 1] = (char)kw_last(&last->kmer);
	for (size_t i = cnt - 2; i >= 1; i--) {                               /* interior nodes, last to first (:493-521) */
		bead_t *b = &c->b[i];
		gnode_t *nd = b->node;
		nd->inEdge = 1;
		symbol += llink(nd, 0) + llink(nd, 1) + llink(nd, 2) + llink(nd, 3);
		if (b->smaller) { nd->l_links = id; nd->twin = (unsigned)(bal_edge + 1); }
		else { nd->l_links = id + (uint32_t)bal_edge; nd->twin = (unsigned)(1 - bal_edge); }
		seq[i - 1] = (char)kw_last(&b->kmer);
	}
	long long cvg = length > 1 ? symbol / (length - 1) * 10 : symbol / length * 10;
	if (cvg > MAX_EDGE_COV) cvg = MAX_EDGE_COV;
	/* output_1edge (output_pregraph.c:83-100) */
	gzprintf(fp, ">length %d,", length);
	gz_kmer(fp, g, &first->kmer);
	gz_kmer(fp, g, &last->kmer);
	gzprintf(fp, "cvg %d, %d\n", (int)cvg, bal_edge);
	for (int i = 0; i < length; i++) {
		gzputc(fp, "ACTG"[(int)seq[i]]);
		if ((i + 1) % 100 == 0) gzputc(fp, '\n');
	}
	gzputc(fp, '\n');
	g->num_ed += (uint64_t)bal_edge;                                      /* the twin takes the next id (:553) */
}

static uint64_t build_edges_sequential(graph_t *g, const char *prefix)
{
	char name[4200];
	snprintf(name, sizeof name, "%s.edge.gz", prefix);
	gzFile fp = gzopen(name, "w");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", name); exit(-1); }
	g->patch_mask = 1023;
	g->patch = (gpatch_t *)calloc(1024, sizeof(gpatch_t));
	g->patch_n = 0;
	g->num_ed = 0;
	chain_t c = {NULL, 0, 0};
	char *seqbuf = NULL;
	size_t seqcap = 0;
	uint64_t plain = 0, extra = 0;
	for (uint64_t i = 0; i < g->n; i++) {
		gnode_t *n = &g->nodes[i];
		if (n->linear || n->deleted) continue;
		const kw_t fw = n->seq, bw = kw_rc(n->seq, g->K);
		for (unsigned b = 0; b < 4; b++) {
			if (!rlink(n, b)) continue;                                   /* live: emit_edge zeroes what it used */
			c.n = 0;
			chain_push(&c, n, fw, 1);
			follow(g, &c, b);
			emit_edge(g, &c, fp, &seqbuf, &seqcap, &plain, &extra);
		}
		for (unsigned b = 0; b < 4; b++) {
			if (!llink(n, b)) continue;
			c.n = 0;
			chain_push(&c, n, bw, 0);
			follow(g, &c, b ^ 2u);
			emit_edge(g, &c, fp, &seqbuf, &seqcap, &plain, &extra);
		}
	}
	printf("%llu (%llu) edges %llu extra nodes\n", (unsigned long long)g->num_ed, (unsigned long long)plain, (unsigned long long)extra);
	gzclose(fp);
	free(c.b);
	free(seqbuf);
	return g->num_ed;
}

/* ---- parallel kmer2edges -------------------------------------------------------------------------------------
 * A chain of linear nodes between two non-linear nodes is one edge, discoverable from either end; the reference
 * emits it from whichever end it visits first and zeroes the link at the other end so that it is not emitted
 * twice.  The walks never change while edges are built (interior nodes keep `linear` and their two links until
 * they are stamped, and nobody walks through a stamped chain again when chains are symmetric), so:
 *   1. dry run, parallel: walk from every live port (node, side, base) of every start-eligible node; record where
 *      the walk ends (node + port), its length, palindrome flag and coverage sum;
 *   2. ordered pass, sequential but O(#ports): replay the reference's visiting order on port flags only, hand out
 *      edge ids (twin = next id), register length-1 edges in the patch table; if any chain turns out not to be
 *      symmetric (walking back from its far port does not return to the near port) give up BEFORE anything has
 *      been modified and let build_edges_sequential do the whole job the reference's way;
 *   3. parallel: re-walk every emitted chain, stamp its interior nodes, zero the two end links, format the record;
 *   4. write the records in id order. */
#include "par.h"
#include "big.h"

typedef struct {
	uint64_t far_node;     /* NO_WALK = port not live / node not eligible */
	uint32_t length;
	uint8_t far_port, bal_edge, emitted;
	uint32_t id;
	char *text;            /* formatted record (phase 3) */
	size_t text_len;
} port_t;
#define NO_WALK (~(uint64_t)0)

typedef struct { graph_t *g; uint64_t *starts; uint64_t nstarts; port_t *ports; } edges_ctx;

static inline int port_live(const gnode_t *n, int p) { return p < 4 ? rlink(n, (unsigned)p) != 0 : llink(n, (unsigned)(p - 4)) != 0; }

static void walk_port(graph_t *g, gnode_t *n, int p, chain_t *c)
{
	c->n = 0;
	if (p < 4) { chain_push(c, n, n->seq, 1); follow(g, c, (unsigned)p); }
	else { chain_push(c, n, kw_rc(n->seq, g->K), 0); follow(g, c, (unsigned)(p - 4) ^ 2u); }
}

static void dry_ports(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	edges_ctx *E = (edges_ctx *)vc;
	graph_t *g = E->g;
	chain_t c = {NULL, 0, 0};
	for (uint64_t s = lo; s < hi; s++) {
		gnode_t *n = &g->nodes[E->starts[s]];
		for (int p = 0; p < 8; p++) {
			port_t *P = &E->ports[s * 8 + p];
			P->far_node = NO_WALK;
			if (!port_live(n, p)) continue;
			walk_port(g, n, p, &c);
			const size_t cnt = c.n;
			const bead_t *last = &c.b[cnt - 1], *second_last = &c.b[cnt - 2];
			P->far_node = (uint64_t)(last->node - g->nodes);
			const unsigned fc = kw_first(&second_last->kmer, g->K);
			P->far_port = (uint8_t)(last->smaller ? 4 + fc : (fc ^ 2u));
			P->length = (uint32_t)(cnt - 1);
			P->bal_edge = 0;
			for (size_t i = 0; i < cnt; i++) {
				kw_t r = kw_rc(c.b[i].kmer, g->K);
				if (!kw_eq(&c.b[cnt - 1 - i].kmer, &r)) { P->bal_edge = 1; break; }
			}
		}
	}
	free(c.b);
}

static void scatter_ports(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	void **a = (void **)vc;
	edges_ctx *E = (edges_ctx *)a[0];
	const uint64_t *rec = (const uint64_t *)a[1];
	const uint64_t *slot_of = (const uint64_t *)a[2];
	for (uint64_t r = lo; r < hi; r++) {
		const uint64_t s = slot_of[rec[r * 17]];
		for (int p = 0; p < 8; p++) {
			port_t *P = &E->ports[s * 8 + p];
			const uint64_t far = rec[r * 17 + 1 + 2 * p], meta = rec[r * 17 + 2 + 2 * p];
			P->far_node = far;
			P->length = (uint32_t)meta;
			P->far_port = (uint8_t)(meta >> 32);
			P->bal_edge = (uint8_t)(meta >> 40);
		}
	}
}

/* Edge records -- what the device's sdt_gpu_build_edges hands over, and what the host path below makes itself: RW = 4 + 2 * kw
 * words per edge, in id order: [0] length | bal_edge << 32, [1] cvg, [2] id, [3] offset of the edge's bases in `bases`
 * (one letter per base: the last base of nodes 1..length), then the oriented first and last k-mer, kw words each, most
 * significant first. */
typedef struct { graph_t *g; edges_ctx *E; uint64_t *emit; uint64_t nemit; uint64_t *rec; const uint64_t *boff; char *bases; } stamp_ctx;

static void stamp_edges(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	stamp_ctx *S = (stamp_ctx *)vc;
	graph_t *g = S->g;
	const int K = g->K;
	chain_t c = {NULL, 0, 0};
	for (uint64_t e = lo; e < hi; e++) {
		const uint64_t pi = S->emit[e];
		port_t *P = &S->E->ports[pi];
		gnode_t *n = &g->nodes[S->E->starts[pi / 8]];
		walk_port(g, n, (int)(pi % 8), &c);
		const size_t cnt = c.n;
		const int length = (int)cnt - 1, bal_edge = P->bal_edge;
		bead_t *first = &c.b[0], *second = &c.b[1], *last = &c.b[cnt - 1], *second_last = &c.b[cnt - 2];
		{   /* the two end links (node2edge.c:387-399); each port belongs to exactly one chain */
			const unsigned fc = kw_first(&second_last->kmer, K);
			if (last->smaller) __sync_fetch_and_and(&last->node->l_links, ~(63u << (6 * fc)));
			else { const uint32_t m = ~(63u << (6 * (fc ^ 2u))); uint32_t *w = (uint32_t *)&last->node->l_links + 1; __sync_fetch_and_and(w, m | 0xFF000000u); }
			const unsigned lc = kw_last(&second->kmer);
			if (first->smaller) { const uint32_t m = ~(63u << (6 * lc)); uint32_t *w = (uint32_t *)&first->node->l_links + 1; __sync_fetch_and_and(w, m | 0xFF000000u); }
			else __sync_fetch_and_and(&first->node->l_links, ~(63u << (6 * (lc ^ 2u))));
		}
		/* interior nodes, last to first: add up the LEFT link counters, then overwrite them with the edge id
		 * (node2edge.c:493-521).  In a palindromic chain a node occurs twice; its second visit then reads the id
		 * it was just stamped with -- upstream behaviour, kept. */
		long long symbol = length == 1 ? (long long)first->node->count : 0;
		for (size_t i = cnt - 2; i >= 1; i--) {
			bead_t *b = &c.b[i];
			gnode_t *nd = b->node;
			nd->inEdge = 1;
			symbol += llink(nd, 0) + llink(nd, 1) + llink(nd, 2) + llink(nd, 3);
			if (b->smaller) { nd->l_links = P->id; nd->twin = (unsigned)(bal_edge + 1); }
			else { nd->l_links = P->id + (uint32_t)bal_edge; nd->twin = (unsigned)(1 - bal_edge); }
		}
		long long cvg = length > 1 ? symbol / (length - 1) * 10 : symbol / length * 10;
		if (cvg > MAX_EDGE_COV) cvg = MAX_EDGE_COV;
		uint64_t *R = S->rec + e * 12;
		R[0] = (uint64_t)length | ((uint64_t)bal_edge << 32);
		R[1] = (uint64_t)cvg;
		R[2] = P->id;
		R[3] = S->boff[e];
		for (int w = 0; w < 4; w++) { R[4 + w] = first->kmer.w[w]; R[8 + w] = last->kmer.w[w]; }
		char *q = S->bases + S->boff[e];
		for (size_t i = 1; i < cnt; i++) q[i - 1] = "ACTG"[kw_last(&c.b[i].kmer)];
	}
	free(c.b);
}

#include <unistd.h>
/* output_1edge (output_pregraph.c:83-100) for a run of edge records, then one gzip member of it: gzopen / gzgets
 * (loadPreGraph.c:439-449) read through member boundaries */
typedef struct { const graph_t *g; const uint64_t *rec; int kw; const char *bases; const uint64_t *cut; unsigned char **out; size_t *len; volatile int failed; } gz_ctx;

static inline size_t fmt_hex(char *p, uint64_t v)
{
	char tmp[16];
	int n = 0;
	do { tmp[n++] = "0123456789abcdef"[v & 15]; v >>= 4; } while (v);
	for (int k = 0; k < n; k++) p[k] = tmp[n - 1 - k];
	return (size_t)n;
}
static inline size_t fmt_dec(char *p, uint64_t v)
{
	char tmp[24];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	for (int k = 0; k < n; k++) p[k] = tmp[n - 1 - k];
	return (size_t)n;
}

static void gz_chunks(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	gz_ctx *Z = (gz_ctx *)vc;
	const int kw = Z->kw, RW = 4 + 2 * kw, nw = Z->g->nw;
	for (uint64_t k = lo; k < hi; k++) {
		size_t cap = 64;
		for (uint64_t e = Z->cut[k]; e < Z->cut[k + 1]; e++) {
			const uint64_t length = Z->rec[e * RW] & 0xFFFFFFFFULL;
			cap += 64 + 34 * (size_t)nw + length + length / 100 + 2;
		}
		char *t = (char *)malloc(cap);
		size_t o = 0;
		for (uint64_t e = Z->cut[k]; e < Z->cut[k + 1]; e++) {
			const uint64_t *R = Z->rec + e * RW;
			const uint64_t length = R[0] & 0xFFFFFFFFULL;
			memcpy(t + o, ">length ", 8); o += 8;
			o += fmt_dec(t + o, length);
			t[o++] = ',';
			for (int end = 0; end < 2; end++) {                             /* print_kmer_gz of the emulated variant (kmer.c:518-545) */
				const uint64_t *w = R + 4 + end * kw;
				if (nw == 1 && !w[kw - 1]) { memcpy(t + o, "0x0,", 4); o += 4; continue; }
				for (int q = 0; q < nw; q++) {
					const int src = q - (nw - kw);                          /* a variant wider than the keys prints zero words in front */
					o += fmt_hex(t + o, src >= 0 ? w[src] : 0);
					t[o++] = q + 1 < nw ? ' ' : ',';
				}
			}
			memcpy(t + o, "cvg ", 4); o += 4;
			o += fmt_dec(t + o, R[1]);
			t[o++] = ','; t[o++] = ' ';
			t[o++] = (char)('0' + ((R[0] >> 32) & 1));
			t[o++] = '\n';
			const char *q = Z->bases + R[3];
			for (uint64_t i = 0; i < length; i += 100) {                     /* a newline after every 100th base and at the end */
				const uint64_t m = length - i < 100 ? length - i : 100;
				memcpy(t + o, q + i, m); o += m;
				if (m == 100) t[o++] = '\n';
			}
			t[o++] = '\n';
		}
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (deflateInit2(&zs, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) { Z->failed = 1; free(t); return; }
		const size_t zcap = deflateBound(&zs, (uLong)o) + 64;
		unsigned char *out = (unsigned char *)malloc(zcap);
		zs.next_out = out;
		zs.avail_out = (uInt)zcap;
		zs.next_in = (Bytef *)t;
		zs.avail_in = (uInt)o;
		if (deflate(&zs, Z_FINISH) == Z_STREAM_ERROR) Z->failed = 1;
		Z->len[k] = zcap - zs.avail_out;
		Z->out[k] = out;
		deflateEnd(&zs);
		free(t);
	}
}

#include <time.h>
static double ed_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
#define EPHASE(name) do { if (sdt_env("SDT_TIMING")) { double t_ = ed_now(); fprintf(stderr, "[edges]    %-26s %9.1f ms\n", name, t_ - t_sub); t_sub = t_; } } while (0)

/* the host's own records: port walks (dry run or the device's), ordered ids, parallel stamping.  Returns 0, or 2 when a chain
 * is not symmetric (nothing has been modified then) */
static int host_edge_records(graph_t *g, uint64_t **rec_out, uint64_t *nrec_out, uint64_t *num_ed_out, char **bases_out, uint64_t *nb_out)
{
	double t_sub = ed_now();
	edges_ctx E;
	E.g = g;
	E.starts = (uint64_t *)malloc((g->n + 1) * sizeof(uint64_t));
	E.nstarts = 0;
	for (uint64_t i = 0; i < g->n; i++)
		if (!g->nodes[i].linear && !g->nodes[i].deleted) E.starts[E.nstarts++] = i;
	E.ports = (port_t *)calloc(E.nstarts * 8 + 8, sizeof(port_t));
	EPHASE("collect starts");
	/* node index -> start slot, for far ends that are start-eligible themselves */
	uint64_t *slot_of = (uint64_t *)malloc((g->n + 1) * sizeof(uint64_t));
	memset(slot_of, 0xFF, (g->n + 1) * sizeof(uint64_t));
	for (uint64_t s = 0; s < E.nstarts; s++) slot_of[E.starts[s]] = s;
	if (g->dev_edge_ports) {
		/* the walks answered by the device mirror of the graph (sdt_gpu_edge_ports), scattered to their start slots */
		uint64_t *rec = NULL, nrec = 0;
		if (g->dev_edge_ports(g, &rec, &nrec) != 0 || nrec != E.nstarts) {
			printf("the device dry run failed (%llu records for %llu start nodes). Now exit to system...\n", (unsigned long long)nrec, (unsigned long long)E.nstarts);
			exit(1);
		}
		graph_clear_dirty(g);          /* the hook brought the mirror up to date */
		g->dn = 0;
		void *sa[3] = {&E, rec, slot_of};
		par_for(0, nrec, 4096, scatter_ports, sa);
		free(rec);
		EPHASE("device walks");
	} else {
		par_for(0, E.nstarts, 256, dry_ports, &E);
		EPHASE("dry walks");
	}
	/* ordered pass on port flags */
	uint8_t *zeroed = (uint8_t *)calloc(g->n + 1, 1);
	uint64_t *emit = (uint64_t *)malloc((E.nstarts * 8 + 8) * sizeof(uint64_t));
	uint64_t *boff = (uint64_t *)malloc((E.nstarts * 8 + 8) * sizeof(uint64_t));
	uint64_t nemit = 0, num_ed = 0, nb = 0;
	int symmetric = 1;
	for (uint64_t s = 0; s < E.nstarts && symmetric; s++) {
		const uint64_t ni = E.starts[s];
		for (int p = 0; p < 8; p++) {
			port_t *P = &E.ports[s * 8 + p];
			if (P->far_node == NO_WALK || (zeroed[ni] >> p) & 1) continue;
			/* symmetric? the far port, if a walk starts there, must come back to this port */
			const uint64_t fs = slot_of[P->far_node];
			if (fs != NO_WALK) {
				const port_t *Q = &E.ports[fs * 8 + P->far_port];
				if (Q->far_node != NO_WALK && !(Q->far_node == ni && Q->far_port == p)) { symmetric = 0; break; }
			}
			P->emitted = 1;
			P->id = (uint32_t)(++num_ed);
			num_ed += P->bal_edge;
			zeroed[ni] |= (uint8_t)(1u << p);
			zeroed[P->far_node] |= (uint8_t)(1u << P->far_port);
			boff[nemit] = nb;
			nb += P->length;
			emit[nemit++] = s * 8 + (uint64_t)p;
		}
	}
	free(zeroed);
	free(slot_of);
	EPHASE("ordered ids");
	if (!symmetric) {
		free(emit); free(boff); free(E.ports); free(E.starts);
		return 2;
	}
	uint64_t *rec = (uint64_t *)malloc((nemit + 1) * 12 * sizeof(uint64_t));
	char *bases = (char *)malloc(nb + 16);
	stamp_ctx S = {g, &E, emit, nemit, rec, boff, bases};
	par_for(0, nemit, 64, stamp_edges, &S);
	EPHASE("stamp + records");
	free(emit); free(boff); free(E.ports); free(E.starts);
	*rec_out = rec; *nrec_out = nemit; *num_ed_out = num_ed; *bases_out = bases; *nb_out = nb;
	return 0;
}

/* <prefix>.edge.gz as a sequence of gzip members, each formatted and deflated by one thread over ~4 MB of records */
typedef struct { graph_t *g; uint64_t *rec; char *bases; int kw; uint64_t nemit; char name[4200]; } ew_job;
static void *edge_writer(void *v)
{
	ew_job *W = (ew_job *)v;
	const graph_t *g = W->g;
	uint64_t *rec = W->rec;
	char *bases = W->bases;
	const int kw = W->kw, RW = 4 + 2 * kw;
	const uint64_t nemit = W->nemit;
	const char *name = W->name;
	const double t0 = ed_now();
	FILE *fz = fopen(name, "wb");
	if (!fz) { printf("Cannot open %s. Now exit to system...\n", name); exit(-1); }
	{
		uint64_t *cut = (uint64_t *)malloc((nemit + 2) * sizeof(uint64_t));
		uint64_t ncut = 0;
		size_t acc = 0;
		const size_t chunk_bytes = sdt_test_env("SDT_GZ_CHUNK") ? (size_t)atol(sdt_test_env("SDT_GZ_CHUNK")) : (4u << 20);   /* the env var is for the tests */
		cut[ncut++] = 0;
		for (uint64_t e = 0; e < nemit; e++) {
			acc += (size_t)(rec[e * RW] & 0xFFFFFFFFULL) + 60;
			if (acc >= chunk_bytes) { cut[ncut++] = e + 1; acc = 0; }
		}
		if (cut[ncut - 1] != nemit) cut[ncut++] = nemit;
		const uint64_t nchunks = ncut - 1;
		gz_ctx Z = {g, rec, kw, bases, cut, (unsigned char **)calloc(nchunks + 1, sizeof(unsigned char *)), (size_t *)calloc(nchunks + 1, sizeof(size_t)), 0};
		par_for(0, nchunks, 1, gz_chunks, &Z);
		if (Z.failed) { printf("deflate failed on %s\n", name); exit(-1); }
		for (uint64_t k = 0; k < nchunks; k++) { fwrite(Z.out[k], 1, Z.len[k], fz); free(Z.out[k]); }
		if (nchunks == 0) {                                  /* no edges: still a valid (empty) gzip file */
			gzFile fe = gzdopen(dup(fileno(fz)), "w1");
			if (fe) gzclose(fe);
		}
		free(Z.out); free(Z.len); free(cut);
	}
	fclose(fz);
	free(rec); free(bases);
	if (sdt_env("SDT_TIMING")) fprintf(stderr, "[edges]    text + gzip write (beside the next phase) %9.1f ms\n", ed_now() - t0);
	free(W);
	return NULL;
}

void graph_edges_join(graph_t *g)
{
	if (g && g->edge_writer_on) { pthread_join(g->edge_writer, NULL); g->edge_writer_on = 0; }
}

uint64_t graph_build_edges(graph_t *g, const char *prefix)
{
	double t_sub = ed_now();
	uint64_t *rec = NULL, nemit = 0, num_ed = 0, nb = 0;
	char *bases = NULL;
	int kw = 4, rc;
	g->edges_on_device = 0;
	if (g->dev_build_edges) {
		/* the whole of kmer2edges from the device mirror (sdt_gpu_build_edges): walks, ordered ids, stamping of the path words
		 * the second read pass reads -- the host nodes are NOT stamped in this mode; the records come back for the writers */
		rc = g->dev_build_edges(g, &rec, &kw, &nemit, &num_ed, &bases, &nb);
		if (rc != 0 && rc != 2) { printf("the device edge builder failed. Now exit to system...\n"); exit(1); }
		if (rc == 0) {
			graph_clear_dirty(g);
			g->dn = 0;
			g->edges_on_device = 1;
		}
		EPHASE("device: walks, ids, stamps");
	} else {
		rc = host_edge_records(g, &rec, &nemit, &num_ed, &bases, &nb);
	}
	if (rc == 2)
		return build_edges_sequential(g, prefix);
	const int RW = 4 + 2 * kw;
	/* length-1 edges: canonical (K+1)-mer -> edge id, in emission order (node2edge.c:404-463) */
	g->patch_mask = 1023;
	g->patch = (gpatch_t *)calloc(1024, sizeof(gpatch_t));
	g->patch_n = 0;
	uint64_t extra = 0;
	for (uint64_t e = 0; e < nemit; e++) {
		const uint64_t *R = rec + e * RW;
		if ((R[0] & 0xFFFFFFFFULL) != 1) continue;
		extra++;
		kw_t from = {{0, 0, 0, 0}}, to = {{0, 0, 0, 0}};
		for (int w = 0; w < kw; w++) { from.w[4 - kw + w] = R[4 + w]; to.w[4 - kw + w] = R[4 + kw + w]; }
		const uint32_t id = (uint32_t)R[2], bal_edge = (uint32_t)(R[0] >> 32) & 1u;
		kw_t plus;
		plus.w[0] = (from.w[0] << 2) | (from.w[1] >> 62);
		plus.w[1] = (from.w[1] << 2) | (from.w[2] >> 62);
		plus.w[2] = (from.w[2] << 2) | (from.w[3] >> 62);
		plus.w[3] = (from.w[3] << 2) | kw_last(&to);
		kw_t bal = kw_rc_kplus1(plus, g->K);
		if (kw_less(&plus, &bal)) patch_put(g, &plus, id, (uint8_t)(bal_edge + 1));
		else patch_put(g, &bal, id + bal_edge, (uint8_t)(1 - bal_edge));
	}
	EPHASE("patch table");
	/* <prefix>.edge.gz is written by a thread of its own, beside the second read pass (the device is busy there, the host is not):
	 * graph_edges_join waits for it */
	{
		ew_job *W = (ew_job *)malloc(sizeof *W);
		W->g = g; W->rec = rec; W->bases = bases; W->kw = kw; W->nemit = nemit;
		snprintf(W->name, sizeof W->name, "%s.edge.gz", prefix);
		if (sdt_tuning_env("SDT_EDGES_INLINE") || pthread_create(&g->edge_writer, NULL, edge_writer, W) != 0) edge_writer(W);
		else g->edge_writer_on = 1;
	}
	EPHASE("text + gzip write (started)");
	g->num_ed = num_ed;
	printf("%llu (%llu) edges %llu extra nodes\n", (unsigned long long)num_ed, (unsigned long long)nemit, (unsigned long long)extra);
	return num_ed;                                           /* (rec and bases belong to the writer now) */
}

/* ---- host stand-in for sdt_gpu_edge_ports (sdt-graphcheck, the CPU tests): the same 17-word records from dry_ports ---- */
void graph_emulate_device_cuts(graph_t *g);

static int emu_edge_ports(graph_t *g, uint64_t **records, uint64_t *n_records)
{
	edges_ctx E;
	E.g = g;
	E.starts = (uint64_t *)malloc((g->n + 1) * sizeof(uint64_t));
	E.nstarts = 0;
	for (uint64_t i = 0; i < g->n; i++)
		if (!g->nodes[i].linear && !g->nodes[i].deleted) E.starts[E.nstarts++] = i;
	E.ports = (port_t *)calloc(E.nstarts * 8 + 8, sizeof(port_t));
	par_for(0, E.nstarts, 256, dry_ports, &E);
	uint64_t *rec = (uint64_t *)malloc((E.nstarts + 1) * 17 * sizeof(uint64_t));
	for (uint64_t s = 0; s < E.nstarts; s++) {
		uint64_t *r = &rec[(E.nstarts - 1 - s) * 17];                      /* any order */
		r[0] = E.starts[s];
		for (int p = 0; p < 8; p++) {
			const port_t *P = &E.ports[s * 8 + p];
			r[1 + 2 * p] = P->far_node;
			r[2 + 2 * p] = P->far_node == NO_WALK ? 0 : ((uint64_t)P->length | ((uint64_t)P->far_port << 32) | ((uint64_t)P->bal_edge << 40));
		}
	}
	*records = rec;
	*n_records = E.nstarts;
	free(E.ports); free(E.starts);
	return 0;
}

void graph_emulate_device(graph_t *g)
{
	graph_emulate_device_cuts(g);
	g->dev_edge_ports = emu_edge_ports;
}
