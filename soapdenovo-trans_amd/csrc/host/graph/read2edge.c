/* read2edge.c -- prlRead2edge: second pass over the reads, read -> path of edge ids -> arcs -> <prefix>.preArc
 * (prlRead2path.c:817-1335; per-read logic parse1read :617-789, search1kmerPlus :575-615, arcs :190-241,415-430,
 * output :454-505).
 *
 * Per read (independent of every other read): look every k-mer up (canonical + which strand), then scan:
 *   - a deleted node, or a linear node that no edge claimed (a floating loop), resets the partial path when fewer
 *     than 2 items were kept, otherwise ends the read.  The "previous vertex" memory is NOT reset by this
 *     (upstream behaviour, :650-663) -- a stale (K+1)-mer may be formed afterwards; it then fails to resolve and
 *     terminates the path, exactly as in the reference;
 *   - a linear node contributes its edge id (l_links, or l_links + twin - 1 on the other strand) unless it equals
 *     the id kept just before;
 *   - two consecutive vertex (non-linear) nodes contribute the canonical (K+1)-mer spanning them, resolved
 *     through the patch table to the id of that length-1 edge; unresolved = 0 = path terminator.
 * Adjacent non-zero ids (a, b) add one to arc a -> b.  The reference keeps, per from-edge, a list with new arcs
 * pushed at the head, and every from-edge is handled by one thread in read order -- so the printed order is
 * "most recent first appearance first".  Here each arc remembers the ordinal of its first appearance instead,
 * which makes accumulation order-free (batches may arrive in any order, e.g. paired files one after the other).
 */
#include "graph.h"
#include <stdlib.h>
#include <string.h>
#include "par.h"
#include "big.h"

typedef struct { uint64_t key; uint64_t first; uint32_t mult; } arc_t;      /* key = from << 32 | to, 0 = empty */

struct arcs {
	arc_t *tab;
	uint64_t mask, n;
	uint64_t reads_dropped;
};

static arc_t *arc_slot(struct arcs *A, uint64_t key)
{
	uint64_t h = (key * 0x9E3779B97F4A7C15ULL) >> 20 & A->mask;
	while (A->tab[h].key && A->tab[h].key != key) h = (h + 1) & A->mask;
	return &A->tab[h];
}

static void arc_add(struct arcs *A, uint32_t from, uint32_t to, uint64_t ord)
{
	if ((A->n + 1) * 2 > A->mask + 1) {
		arc_t *old = A->tab;
		const uint64_t cap = A->mask + 1;
		A->mask = cap * 2 - 1;
		A->tab = (arc_t *)calloc(cap * 2, sizeof(arc_t));
		for (uint64_t i = 0; i < cap; i++)
			if (old[i].key) *arc_slot(A, old[i].key) = old[i];
		free(old);
	}
	const uint64_t key = ((uint64_t)from << 32) | to;
	arc_t *s = arc_slot(A, key);
	if (!s->key) {
		s->key = key; s->first = ord; s->mult = 1;           /* prlAllocatePreArc: multiplicity starts at 1 */
		A->n++;
	} else {
		s->mult++;
		if (ord < s->first) s->first = ord;
	}
}

struct arcs *arcs_new(void)
{
	struct arcs *A = (struct arcs *)calloc(1, sizeof *A);
	A->mask = (1u << 16) - 1;
	A->tab = (arc_t *)calloc(A->mask + 1, sizeof(arc_t));
	return A;
}

void arcs_free(struct arcs *A)
{
	if (!A) return;
	free(A->tab);
	free(A);
}

/* one read given as codes 0..3; ordinal = its position in the reference's read stream */
void arcs_add_read(graph_t *g, struct arcs *A, const uint8_t *codes, int len, uint64_t ordinal)
{
	const int K = g->K;
	if (len < K + 1) return;                                               /* prlRead2path.c:969,1052,1116,1196 */
	const int n = len - K + 1;
	uint64_t stackbuf[512];
	uint64_t *mix = n <= 512 ? stackbuf : (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
	kw_t word = {{0, 0, 0, 0}};
	for (int i = 0; i < K - 1; i++) word = kw_next(word, codes[i], K);
	int retain = 0, pos = 0, have_prev = 0;
	kw_t prev_kmer = {{0, 0, 0, 0}};
	for (int j = 0; j < n; j++) {
		word = kw_next(word, codes[j + K - 1], K);
		int smaller;
		gnode_t *nd = graph_find_oriented(g, word, &smaller);
		if (nd->deleted || (nd->linear && !nd->inEdge)) {
			if (retain < 2) { retain = 0; pos = 0; }
			else break;
			continue;
		}
		if (nd->linear) {
			const uint64_t id = smaller ? nd->l_links : nd->l_links + nd->twin - 1;
			if (retain == 0 || have_prev) {
				retain++;
				mix[pos++] = id;
				have_prev = 0;
			} else if (id != mix[pos - 1]) {
				retain++;
				mix[pos++] = id;
			}
		} else {
			/* `word` is the oriented k-mer of this vertex: seq when smaller, its reverse complement otherwise */
			if (have_prev) {
				retain++;
				kw_t plus;
				plus.w[0] = (prev_kmer.w[0] << 2) | (prev_kmer.w[1] >> 62);
				plus.w[1] = (prev_kmer.w[1] << 2) | (prev_kmer.w[2] >> 62);
				plus.w[2] = (prev_kmer.w[2] << 2) | (prev_kmer.w[3] >> 62);
				plus.w[3] = (prev_kmer.w[3] << 2) | kw_last(&word);
				kw_t bal = kw_rc_kplus1(plus, K);
				const int plus_smaller = kw_less(&plus, &bal);
				const gpatch_t *p = graph_find_patch(g, plus_smaller ? &plus : &bal);
				mix[pos++] = !p ? 0 : (plus_smaller ? p->edge : (uint64_t)p->edge + p->twin - 1);
			}
			have_prev = 1;
			prev_kmer = word;
		}
	}
	if (retain < 1) A->reads_dropped++;
	if (retain >= 2) {
		/* signal 6 (:190-241): stop at the first unresolved item */
		for (int j = 0; j + 1 < pos; j++) {
			if (mix[j] == 0 || mix[j + 1] == 0) break;
			arc_add(A, (uint32_t)mix[j], (uint32_t)mix[j + 1], (ordinal << 16) | (uint64_t)j);
		}
	}
	if (mix != stackbuf) free(mix);
}

static int cmp_arc(const void *a, const void *b)
{
	const arc_t *x = (const arc_t *)a, *y = (const arc_t *)b;
	const uint32_t fx = (uint32_t)(x->key >> 32), fy = (uint32_t)(y->key >> 32);
	if (fx != fy) return fx < fy ? -1 : 1;
	return x->first > y->first ? -1 : x->first < y->first;                  /* most recent first appearance first */
}

/* (from ascending, first appearance descending): LSD radix sort on the composite key from << 64 | ~first, 11 bits per pass,
 * skipping the digits that are the same in every key */
static void sort_arcs(arc_t *v, uint64_t m)
{
	if (m < 64) { qsort(v, m, sizeof(arc_t), cmp_arc); return; }
	{
		/* (the device hands its arcs over in this order already) */
		uint64_t i = 1;
		while (i < m && cmp_arc(&v[i - 1], &v[i]) <= 0) i++;
		if (i == m) return;
	}
	arc_t *tmp = (arc_t *)malloc(m * sizeof(arc_t));
	if (!tmp) { qsort(v, m, sizeof(arc_t), cmp_arc); return; }
	uint64_t or_lo = 0, and_lo = ~0ULL, or_hi = 0, and_hi = ~0ULL;
	for (uint64_t i = 0; i < m; i++) {
		const uint64_t lo = ~v[i].first, hi = v[i].key >> 32;
		or_lo |= lo; and_lo &= lo; or_hi |= hi; and_hi &= hi;
	}
	arc_t *src = v, *dst = tmp;
	for (int pass = 0; pass < 9; pass++) {                              /* digits 0..5: ~first, 6..8: from */
		const int in_hi = pass >= 6, shift = in_hi ? (pass - 6) * 11 : pass * 11;
		const uint64_t varying = in_hi ? (or_hi ^ and_hi) : (or_lo ^ and_lo);
		if (!((varying >> shift) & 0x7FF)) continue;
		uint64_t *cnt = (uint64_t *)calloc(2049, sizeof(uint64_t));
		for (uint64_t i = 0; i < m; i++) cnt[(((in_hi ? src[i].key >> 32 : ~src[i].first) >> shift) & 0x7FF) + 1]++;
		for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
		for (uint64_t i = 0; i < m; i++) dst[cnt[((in_hi ? src[i].key >> 32 : ~src[i].first) >> shift) & 0x7FF]++] = src[i];
		free(cnt);
		arc_t *t = src; src = dst; dst = t;
	}
	if (src != v) memcpy(v, src, m * sizeof(arc_t));
	free(tmp);
}

static inline size_t arc_dec(char *p, uint32_t v)
{
	char tmp[12];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	for (int k = 0; k < n; k++) p[k] = tmp[n - 1 - k];
	return (size_t)n;
}

/* output_arcs (:454-505) */
int arcs_write(struct arcs *A, const char *prefix)
{
	char name[4200];
	snprintf(name, sizeof name, "%s.preArc", prefix);
	FILE *fp = fopen(name, "w");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", name); exit(-1); }
	arc_t *v = (arc_t *)malloc((A->n + 1) * sizeof(arc_t));
	uint64_t m = 0;
	for (uint64_t i = 0; i <= A->mask; i++)
		if (A->tab[i].key) v[m++] = A->tab[i];
	sort_arcs(v, m);
	const size_t cap = (size_t)1 << 20;
	char *buf = (char *)malloc(cap + 64);
	size_t o = 0;
	for (uint64_t i = 0; i < m;) {
		const uint32_t from = (uint32_t)(v[i].key >> 32);
		o += arc_dec(buf + o, from);
		for (; i < m && (uint32_t)(v[i].key >> 32) == from; i++) {
			buf[o++] = ' ';
			o += arc_dec(buf + o, (uint32_t)v[i].key);
			buf[o++] = ' ';
			o += arc_dec(buf + o, v[i].mult);
			if (o >= cap) { fwrite(buf, 1, o, fp); o = 0; }
		}
		buf[o++] = '\n';
		if (o >= cap) { fwrite(buf, 1, o, fp); o = 0; }
	}
	fwrite(buf, 1, o, fp);
	free(buf);
	fclose(fp);
	free(v);
	return 0;
}

/* same file from arcs accumulated elsewhere (the GPU second pass): arrays of (from, to, multiplicity, first) */
int arcs_write_arrays(const char *prefix, const uint32_t *from, const uint32_t *to, const uint32_t *mult,
                      const uint64_t *first, uint64_t n)
{
	struct arcs A;
	A.tab = (arc_t *)calloc(n + 1, sizeof(arc_t));
	A.mask = n;             /* arcs_write scans 0..mask */
	A.n = n;
	for (uint64_t i = 0; i < n; i++) {
		A.tab[i].key = ((uint64_t)from[i] << 32) | to[i];
		A.tab[i].first = first[i];
		A.tab[i].mult = mult[i];
	}
	const int rc = arcs_write(&A, prefix);
	free(A.tab);
	return rc;
}
