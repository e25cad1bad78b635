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
	while (o->linear) {
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
		kw_t bal = kw_rc(plus, K + 1);
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

uint64_t graph_build_edges(graph_t *g, const char *prefix)
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
