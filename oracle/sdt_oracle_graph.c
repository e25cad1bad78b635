/*
 * sdt_oracle_graph.c -- CPU restatement of the k-mer-graph cleaning passes of `pregraph`
 * (cutTipPreGraph.c: removeMinorOut / clipKmerFromNode, removeSingleTips / removeMinorTips / clipTipFromNode,
 * Mark1in1outNode), of kmer2edges (node2edge.c), of prlRead2edge (prlRead2path.c) and of output_vertex.  TEST INFRASTRUCTURE ONLY (see sdt_oracle.h).
 *
 * These passes are ORDER DEPENDENT (survey 7.3-1): they visit set 0..p-1, slot 0..size-1 of the reference's
 * table layout and mutate neighbours as they go.  The oracle's sets have that exact layout (sdto_set_put is a
 * bit-exact put_kmerset), so running the passes here reproduces the reference's result for the same -p.
 * Pinned by tests/test_oracle_vs_reference.py::test_case_vertex against the reference's *.vertex files; kmer2edges
 * (node2edge.c) is restated here too: ::test_case_edge_file (sdto_write_edges: the reference's *.edge.gz byte for byte) and
 * ::test_case_edges_from_port_walks (its read-only walk, sdto_edge_port, which the device dry run is compared with); and so is
 * the second read pass, prlRead2edge (prlRead2path.c): ::test_case_prearc (sdto_read2edge: *.preArc byte for byte).
 */
#include "sdt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FL_NULL(f, i) (((f)[(i) >> 4] >> (((i) & 0x0f) << 1)) & 0x01)

static inline uint32_t lcov(const sdto_node *n, int b) { return (n->l_links >> (6 * b)) & 0x3f; }
static inline uint32_t rcov(const sdto_node *n, int b) { return (n->r_links >> (6 * b)) & 0x3f; }
static inline void lzero(sdto_node *n, int b) { n->l_links &= ~(0x3fu << (6 * b)); }
static inline void rzero(sdto_node *n, int b) { n->r_links &= ~(0x3fu << (6 * b)); }

/* newhash.c:510-538 */
static int branch2prev(const sdto_node *n) { int c = 0; for (int b = 0; b < 4; b++) c += lcov(n, b) > 0; return c; }
static int branch2next(const sdto_node *n) { int c = 0; for (int b = 0; b < 4; b++) c += rcov(n, b) > 0; return c; }

/* newhash.c:540-562: orientation-aware removal of one link */
static void dislink2prev(sdto_node *n, int ch, int smaller) { if (smaller) lzero(n, ch); else rzero(n, ch ^ 2); }
static void dislink2next(sdto_node *n, int ch, int smaller) { if (smaller) rzero(n, ch); else lzero(n, ch ^ 2); }

/* the lookup every pass uses: canonicalise (KmerLarger(word, bal) -> take bal, smaller = 0), pick the set by
 * hash_kmer % thrd_num, search_kmerset; a miss is fatal in the reference (exit(1), cutTipPreGraph.c:124-140) */
static sdto_node *find_oriented(sdto_sets *S, sdto_kmer word, int *smaller, sdto_kmer *canon_out)
{
	sdto_kmer bal = sdto_reverse_complement(word, S->K);
	sdto_kmer canon = word;
	*smaller = 1;
	if (sdto_kmer_smaller(bal, word)) {          /* KmerLarger(word, bal_word) */
		canon = bal;
		*smaller = 0;
	}
	if (canon_out) *canon_out = canon;
	sdto_set *set = S->sets[sdto_hash_kmer(canon, S->nw) % (uint64_t)S->nsets];
	uint64_t slot;
	if (!sdto_set_search(set, canon, S->nw, &slot)) {
		fprintf(stderr, "oracle: kmer %llx %llx %llx %llx not found\n", (unsigned long long)canon.w[0],
		        (unsigned long long)canon.w[1], (unsigned long long)canon.w[2], (unsigned long long)canon.w[3]);
		abort();
	}
	return set->array + slot;
}

/* cutTipPreGraph.c:1121-1229 thread_mark + Mark1in1outNode: only ever SETS linear, skips deleted / already linear */
static uint64_t mark_more_linear(sdto_sets *S)
{
	uint64_t n = 0;
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			sdto_node *rs = s->array + i;
			if (rs->deleted || rs->linear) continue;
			if (branch2prev(rs) == 1 && branch2next(rs) == 1) {
				rs->linear = 1;
				n++;
			}
		}
	}
	return n;
}

/* ---- clipKmerFromNode (cutTipPreGraph.c:591-1010) ---- */

/* after q was judged a minor branch: q.deleted = 1 and every neighbour of q forgets its link to q
 * (:685-820 and the mirrored :886-1005).  q's own links stay. */
static void cut_out_node(sdto_sets *S, sdto_node *q)
{
	const sdto_kmer qseq = q->seq;
	int K = S->K, smaller;
	q->deleted = 1;
	for (int ch = 0; ch < 4; ch++) {
		if (!lcov(q, ch)) continue;
		sdto_node *x = find_oriented(S, sdto_prev_kmer(qseq, ch, K), &smaller, NULL);
		dislink2next(x, sdto_last_char(qseq), smaller);
		x->linear = (branch2prev(x) == 1 && branch2next(x) == 1);
	}
	for (int ch = 0; ch < 4; ch++) {
		if (!rcov(q, ch)) continue;
		sdto_node *y = find_oriented(S, sdto_next_kmer(qseq, ch, K), &smaller, NULL);
		dislink2prev(y, sdto_first_char(qseq, K), smaller);
		y->linear = (branch2prev(y) == 1 && branch2next(y) == 1);
	}
}

static void clip_kmer_from_node(sdto_sets *S, sdto_node *n1, double threshold, uint64_t *kmers_c)
{
	int K = S->K, smaller;
	if (n1->linear || n1->deleted) return;
	int in_num = branch2prev(n1), out_num = branch2next(n1);
	if (in_num <= 1 && out_num <= 1) return;
	if (in_num > 1) {
		/* getmaxofprev (:439-513): max unsaturated count over the linked predecessors */
		int maxIn = 0;
		for (int c = 0; c < 4; c++)
			if (lcov(n1, c)) {
				sdto_node *p = find_oriented(S, sdto_prev_kmer(n1->seq, c, K), &smaller, NULL);
				if ((int)p->count > maxIn) maxIn = (int)p->count;
			}
		if (maxIn) {
			for (int c = 0; c < 4; c++) {
				if (!lcov(n1, c)) continue;        /* read live: earlier cuts may have removed it (:626) */
				sdto_node *p = find_oriented(S, sdto_prev_kmer(n1->seq, c, K), &smaller, NULL);
				int temp = (int)p->count;
				if (temp && (double)temp / maxIn < threshold) {
					(*kmers_c)++;
					cut_out_node(S, p);
				}
			}
		}
	}
	if (out_num > 1) {                             /* out_num was sampled before the in-side cuts (:617,826) */
		int maxOut = 0;
		for (int c = 0; c < 4; c++)
			if (rcov(n1, c)) {
				sdto_node *p = find_oriented(S, sdto_next_kmer(n1->seq, c, K), &smaller, NULL);
				if ((int)p->count > maxOut) maxOut = (int)p->count;
			}
		if (maxOut) {
			for (int c = 0; c < 4; c++) {
				if (!rcov(n1, c)) continue;
				sdto_node *p = find_oriented(S, sdto_next_kmer(n1->seq, c, K), &smaller, NULL);
				int temp = (int)p->count;
				if (temp && (double)temp / maxOut < threshold) {
					(*kmers_c)++;
					cut_out_node(S, p);
				}
			}
		}
	}
}

uint64_t sdto_remove_minor_out(sdto_sets *S, int dd, uint64_t *more_linear)
{
	double threshold = (double)dd / 100;          /* :1014 */
	uint64_t kmers_c = 0;
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++)
			if (!FL_NULL(s->flags, i))
				clip_kmer_from_node(S, s->array + i, threshold, &kmers_c);
	}
	uint64_t ml = mark_more_linear(S);
	if (more_linear) *more_linear = ml;
	return kmers_c;
}

/* ---- clipTipFromNode (cutTipPreGraph.c:43-337) ---- */

/* its read-only first half (:43-281): from a dead end n1 along linear nodes to the node `o` the chain runs into.
 * Returns 0 when there is nothing to decide (n1 linear / deleted / not a dead end / thin and not single / chain longer
 * than cut_len).  pre_word = the oriented k-mer the walk arrived from, smaller = strand on which `o` was reached,
 * thin_stop = the walk ended at a linear node that is not single (:163-166). */
static int tip_walk(sdto_sets *S, const sdto_node *n1, int cut_len, int thin, sdto_node **end, sdto_kmer *pre_out, int *smaller_out,
                    int *thin_stop)
{
	int K = S->K, smaller, ch1, ch;
	sdto_kmer pre_word, word;
	*thin_stop = 0;
	if (n1->linear || n1->deleted) return 0;
	if (thin && !n1->single) return 0;
	int in_num = branch2prev(n1), out_num = branch2next(n1);
	if (in_num == 0 && out_num == 1) {
		pre_word = n1->seq;
		for (ch1 = 0; ch1 < 4; ch1++) if (rcov(n1, ch1)) break;
		word = sdto_next_kmer(pre_word, ch1, K);
	} else if (in_num == 1 && out_num == 0) {
		pre_word = sdto_reverse_complement(n1->seq, K);
		for (ch1 = 0; ch1 < 4; ch1++) if (lcov(n1, ch1)) break;
		word = sdto_next_kmer(pre_word, ch1 ^ 2, K);
	} else {
		return 0;
	}
	int count = 1;
	sdto_kmer canon;
	sdto_node *o = find_oriented(S, word, &smaller, &canon);
	/* from here `word` is the canonical form and `bal` its reverse complement, as in the reference */
	while (o->linear) {
		count++;
		if (thin && !o->single) { *thin_stop = 1; break; }
		if (count > cut_len) return 0;
		if (smaller) {
			pre_word = canon;                                         /* oriented = canonical */
			for (ch = 0; ch < 4; ch++) if (rcov(o, ch)) break;
			word = sdto_next_kmer(pre_word, ch, K);
		} else {
			pre_word = sdto_reverse_complement(canon, K);             /* oriented = the larger strand */
			for (ch = 0; ch < 4; ch++) if (lcov(o, ch)) break;
			word = sdto_next_kmer(pre_word, ch ^ 2, K);
		}
		o = find_oriented(S, word, &smaller, &canon);
	}
	*end = o;
	*pre_out = pre_word;
	*smaller_out = smaller;
	return 1;
}

static int clip_tip_from_node(sdto_sets *S, sdto_node *n1, int cut_len, int thin, uint64_t *tip_c)
{
	int K = S->K, smaller, ch, thin_stop;
	sdto_kmer pre_word;
	sdto_node *o;
	if (!tip_walk(S, n1, cut_len, thin, &o, &pre_word, &smaller, &thin_stop))
		return 0;
	if (branch2next(o) + branch2prev(o) == 1) {                        /* :282-288 isolated path */
		(*tip_c)++;
		n1->deleted = 1;
		o->deleted = 1;
		return 1;
	}
	ch = sdto_first_char(pre_word, K);
	if (thin) {                                                        /* :293-300 */
		(*tip_c)++;
		n1->deleted = 1;
		dislink2prev(o, ch, smaller);
		o->linear = 0;
		return 1;
	}
	uint32_t max_links = 0;                                            /* :303-322 */
	for (int c = 0; c < 4; c++) {
		uint32_t v = smaller ? lcov(o, c) : rcov(o, c);
		if (v > max_links) max_links = v;
	}
	uint32_t mine = smaller ? lcov(o, ch) : rcov(o, ch ^ 2);
	if (mine < max_links) {                                            /* :324-349 strict < */
		(*tip_c)++;
		n1->deleted = 1;
		dislink2prev(o, ch, smaller);
		if (branch2prev(o) == 1 && branch2next(o) == 1)
			o->linear = 1;
		return 1;
	}
	return 0;
}

/* ---- read-only probes of the passes above, for pinning the device dry runs (tests/test_gpu_parity.py) ---------------
 * The same code paths the ordered passes take (tip_walk is literally clipTipFromNode's first half), minus the writes. */
static sdto_node *node_of(sdto_sets *S, const uint64_t key4[4])
{
	sdto_kmer k;
	memcpy(k.w, key4, sizeof k.w);
	int sm;
	return find_oriented(S, k, &sm, NULL);           /* keys handed in are canonical: sm == 1 */
}

/* overwrite a node's links and flags (bit 0 linear, bit 1 deleted): what a host "write" does to the graph */
void sdto_node_set(sdto_sets *S, const uint64_t key4[4], uint32_t l_links, uint32_t r_links, int linear, int deleted)
{
	sdto_node *n = node_of(S, key4);
	n->l_links = l_links & 0xFFFFFFu;
	n->r_links = r_links & 0xFFFFFFu;
	n->linear = linear != 0;
	n->deleted = deleted != 0;
}

/* clipTipFromNode's walk from one node: 1 = there is a decision to take at end_key4, info = ch | sm << 2 | thin_stop << 3 */
int sdto_tip_walk(sdto_sets *S, const uint64_t key4[4], int cut_len, int thin, uint64_t end_key4[4], int *info)
{
	sdto_node *o;
	sdto_kmer pre;
	int sm, ts;
	if (!tip_walk(S, node_of(S, key4), cut_len, thin, &o, &pre, &sm, &ts))
		return 0;
	memcpy(end_key4, o->seq.w, sizeof o->seq.w);
	*info = sdto_first_char(pre, S->K) | (sm << 2) | (ts << 3);
	return 1;
}

/* the eight neighbours of a node (left links 0..3, then right links 0..3): state[i] = -1 no link, else the strand flag
 * `smaller` of the look-up; nb_key4[i] = the neighbour's canonical k-mer */
void sdto_neighbours(sdto_sets *S, const uint64_t key4[4], uint64_t nb_key4[8][4], int state[8])
{
	const sdto_node *n1 = node_of(S, key4);
	for (int i = 0; i < 8; i++) {
		const int c = i & 3, left = i < 4;
		state[i] = -1;
		if (!(left ? lcov(n1, c) : rcov(n1, c))) continue;
		int sm;
		sdto_node *p = find_oriented(S, left ? sdto_prev_kmer(n1->seq, c, S->K) : sdto_next_kmer(n1->seq, c, S->K), &sm, NULL);
		memcpy(nb_key4[i], p->seq.w, sizeof p->seq.w);
		state[i] = sm;
	}
}

/* kmer2edges' walk from one port of a node that is neither linear nor deleted (startEdgeFromNode + stringBeads +
 * check_iden_kmerList, node2edge.c:58-191,193-310,563-588): port 0..3 = right link ch on the stored strand, port 4..7 =
 * left link ch - 4, walked on the reverse strand with int_comp(ch).  Returns -1 for a node that starts no edge, 0 for a
 * port without a link, 1 otherwise with far_key4 = the first non-linear node, info[0] = the port of the far node the
 * chain arrives through, info[1] = edges' length in k-mer steps (node_c - 1), info[2] = bal_edge (0: the list of
 * oriented k-mers equals the list of their reverse complements read backwards, i.e. the edge is its own twin). */
int sdto_edge_port(sdto_sets *S, const uint64_t key4[4], int port, uint64_t far_key4[4], int info[3])
{
	const sdto_node *n1 = node_of(S, key4);
	const int K = S->K;
	if (n1->linear || n1->deleted) return -1;                      /* :201-204 */
	const int ch1 = port & 3, right = port < 4;
	if (!(right ? rcov(n1, ch1) : lcov(n1, ch1))) return 0;        /* :213-219 / :262-268 */
	size_t cap = 64, n = 0;
	sdto_kmer *list = (sdto_kmer *)malloc(cap * sizeof *list);     /* nodeStack: the oriented k-mer of every bead */
	sdto_kmer oriented = right ? n1->seq : sdto_reverse_complement(n1->seq, K);
	list[n++] = oriented;
	/* stringBeads :58-191 */
	sdto_kmer word = sdto_next_kmer(oriented, right ? ch1 : (ch1 ^ 2), K), canon;
	int smaller, ch;
	sdto_node *o = find_oriented(S, word, &smaller, &canon);
	while (o->linear) {
		if (n + 2 > cap) { cap *= 2; list = (sdto_kmer *)realloc(list, cap * sizeof *list); }
		oriented = smaller ? canon : sdto_reverse_complement(canon, K);
		list[n++] = oriented;
		if (smaller) {
			for (ch = 0; ch < 4; ch++) if (rcov(o, ch)) break;
			word = sdto_next_kmer(oriented, ch, K);
		} else {
			for (ch = 0; ch < 4; ch++) if (lcov(o, ch)) break;
			word = sdto_next_kmer(oriented, ch ^ 2, K);
		}
		o = find_oriented(S, word, &smaller, &canon);
	}
	if (n + 1 > cap) { cap += 1; list = (sdto_kmer *)realloc(list, cap * sizeof *list); }
	list[n++] = smaller ? canon : sdto_reverse_complement(canon, K);
	/* check_iden_kmerList :563-588 on nodeStack and its reverse-complemented copy popped in step */
	int palindrome = 1;
	for (size_t j = 0; j < n && palindrome; j++) {
		const sdto_kmer a = list[n - 1 - j], b = sdto_reverse_complement(list[j], K);
		if (memcmp(a.w, b.w, sizeof a.w) != 0) palindrome = 0;
	}
	const int fc = sdto_first_char(list[n - 2], K);                /* the base the chain leaves behind when it enters the far node */
	memcpy(far_key4, o->seq.w, sizeof o->seq.w);
	info[0] = smaller ? 4 + fc : (fc ^ 2);
	info[1] = (int)(n - 1);
	info[2] = palindrome ? 0 : 1;
	free(list);
	return 1;
}

/* clipKmerFromNode's tests (:591-1010) on the graph as it is: cut[i] = 1 when neighbour i (order as above) would be
 * cut by the ratio test of this junction; returns how many */
int sdto_minor_out_probe(sdto_sets *S, const uint64_t key4[4], double threshold, int cut[8])
{
	const sdto_node *n1 = node_of(S, key4);
	int K = S->K, smaller, ncut = 0;
	for (int i = 0; i < 8; i++) cut[i] = 0;
	if (n1->linear || n1->deleted) return 0;
	const int in_num = branch2prev(n1), out_num = branch2next(n1);
	for (int side = 0; side < 2; side++) {
		if ((side == 0 ? in_num : out_num) <= 1) continue;
		int mx = 0;
		for (int c = 0; c < 4; c++)
			if (side == 0 ? lcov(n1, c) : rcov(n1, c)) {
				sdto_node *p = find_oriented(S, side == 0 ? sdto_prev_kmer(n1->seq, c, K) : sdto_next_kmer(n1->seq, c, K), &smaller, NULL);
				if ((int)p->count > mx) mx = (int)p->count;
			}
		if (!mx) continue;
		for (int c = 0; c < 4; c++) {
			if (!(side == 0 ? lcov(n1, c) : rcov(n1, c))) continue;
			sdto_node *p = find_oriented(S, side == 0 ? sdto_prev_kmer(n1->seq, c, K) : sdto_next_kmer(n1->seq, c, K), &smaller, NULL);
			const int temp = (int)p->count;
			if (temp && (double)temp / mx < threshold) {
				cut[side * 4 + c] = 1;
				ncut++;
			}
		}
	}
	return ncut;
}

uint64_t sdto_remove_single_tips(sdto_sets *S, uint64_t *more_linear)
{
	uint64_t tip_c = 0;
	int cut_len = 2 * S->K;
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++)
			if (!FL_NULL(s->flags, i))
				clip_tip_from_node(S, s->array + i, cut_len, 1, &tip_c);
	}
	uint64_t ml = mark_more_linear(S);
	if (more_linear) *more_linear = ml;
	return tip_c;
}

uint64_t sdto_remove_minor_tips(sdto_sets *S, uint64_t *more_linear)
{
	uint64_t tip_c = 0;
	int cut_len = 2 * S->K;
	for (int t = 0; t < S->nsets; t++) {
		sdto_set *s = S->sets[t];
		int flag = 1;
		while (flag) {                         /* fixed point PER SET before the next set (:385-408) */
			flag = 0;
			for (uint64_t i = 0; i < s->size; i++)
				if (!FL_NULL(s->flags, i))
					flag += clip_tip_from_node(S, s->array + i, cut_len, 0, &tip_c);
		}
	}
	uint64_t ml = mark_more_linear(S);
	if (more_linear) *more_linear = ml;
	return tip_c;
}

/* output_vertex (output_pregraph.c:29-81): every !linear && !deleted node in table order, 8 per line,
 * print_kmer format of the variant (kmer.c:499-516): MER31 "%llx" with 0 printed as "0x0"; MER63 two words;
 * MER127 four words */
/* ---- kmer2edges (node2edge.c:46-588, output_pregraph.c:83-100): the whole pass, in the reference's order, MUTATING the
 * graph as the reference does (end links zeroed so that the twin walk does not emit the edge again, interior nodes' l_links
 * overwritten with the edge id -- which the coverage of a self-complementary chain then reads back, :497-507) and
 * writing the text of <prefix>.edge.gz (uncompressed).  Pinned byte for byte by tests/test_oracle_vs_reference.py::
 * test_case_edge_file.  The (K+1)-mer patch table of length-1 edges (:404-463) is only counted ("extra nodes"). */
typedef struct { sdto_node *node; sdto_kmer kmer; int smaller; } sdto_bead;

/* what kmer2edges leaves in a node for the second read pass: kmer_t.twin (2 bits) and .inEdge, kept in sdto_node.pad */
#define ND_TWIN(n) ((n)->pad & 3)
#define ND_INEDGE(n) (((n)->pad >> 2) & 1)
static void nd_set_edge(sdto_node *n, uint32_t l_links, int twin) { n->l_links = l_links; n->pad = (uint8_t)((twin & 3) | 4); }

/* KmerSetsPatch as one open-addressing table over the 4-word (K+1)-mer (the reference's per-thread sets only decide who looks) */
typedef struct { sdto_kmer key; uint32_t edge; uint8_t twin, used; } sdto_patch;
static uint64_t patch_hash(const sdto_kmer *k)
{
	uint64_t h = 0x9E3779B97F4A7C15ULL;
	for (int i = 0; i < 4; i++) { h ^= k->w[i]; h *= 0xD6E8FEB86659FD93ULL; h ^= h >> 32; }
	return h;
}
static sdto_patch *patch_find(const sdto_sets *S, const sdto_kmer *k)
{
	sdto_patch *t = (sdto_patch *)S->patch;
	if (!t) return NULL;
	for (uint64_t h = patch_hash(k) & (S->patch_cap - 1);; h = (h + 1) & (S->patch_cap - 1)) {
		if (!t[h].used) return NULL;
		if (memcmp(t[h].key.w, k->w, sizeof k->w) == 0) return &t[h];
	}
}
static void patch_put(sdto_sets *S, const sdto_kmer *k, uint32_t edge, int twin)
{
	if (!S->patch || (S->patch_n + 1) * 2 > S->patch_cap) {
		const uint64_t ncap = S->patch_cap ? S->patch_cap * 2 : 1024;
		sdto_patch *old = (sdto_patch *)S->patch, *nt = (sdto_patch *)calloc(ncap, sizeof *nt);
		const uint64_t ocap = S->patch_cap;
		S->patch = nt;
		S->patch_cap = ncap;
		for (uint64_t i = 0; i < ocap; i++)
			if (old[i].used) {
				uint64_t h = patch_hash(&old[i].key) & (ncap - 1);
				while (nt[h].used) h = (h + 1) & (ncap - 1);
				nt[h] = old[i];
			}
		free(old);
	}
	sdto_patch *t = (sdto_patch *)S->patch;
	uint64_t h = patch_hash(k) & (S->patch_cap - 1);
	while (t[h].used && memcmp(t[h].key.w, k->w, sizeof k->w) != 0) h = (h + 1) & (S->patch_cap - 1);
	if (!t[h].used) S->patch_n++;                                                         /* (an existing node is overwritten, :426-443) */
	t[h].key = *k; t[h].edge = edge; t[h].twin = (uint8_t)twin; t[h].used = 1;
}

static uint32_t left_covs(const sdto_node *n) { return lcov(n, 0) + lcov(n, 1) + lcov(n, 2) + lcov(n, 3); }

static void print_kmer_sep(FILE *fp, const sdto_sets *S, sdto_kmer k, char c)       /* kmer.c:517-535 */
{
	const uint64_t *w = k.w;
	if (S->nw == 4)
		fprintf(fp, "%llx %llx %llx %llx", (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3]);
	else if (S->nw == 2)
		fprintf(fp, "%llx %llx", (unsigned long long)w[2], (unsigned long long)w[3]);
	else if (w[3])
		fprintf(fp, "%llx", (unsigned long long)w[3]);
	else
		fprintf(fp, "0x0");
	fputc(c, fp);
}

/* merge_linearV2 :352-560 on the beads b[0..n) of one chain */
static void merge_linear(sdto_sets *S, FILE *fp, sdto_bead *b, int n, int bal_edge, long long *edge_c, uint64_t *edge_counter,
                         uint64_t *extra_nodes, char **seqbuf, size_t *seqcap)
{
	const int K = S->K, length = n - 1;
	if ((size_t)length + 1 > *seqcap) { *seqcap = (size_t)length * 2 + 64; *seqbuf = (char *)realloc(*seqbuf, *seqcap); }
	char *seq = *seqbuf;
	int ci = length - 1;
	sdto_bead *last = &b[n - 1], *second_last = &b[n - 2], *first = &b[0], *second = &b[1];
	seq[ci--] = (char)sdto_last_char(last->kmer);
	dislink2prev(last->node, sdto_first_char(second_last->kmer, K), last->smaller);           /* :382 */
	dislink2next(first->node, sdto_last_char(second->kmer), first->smaller);                  /* :392 */
	long long symbol = 0;
	if (length == 1) {
		(*extra_nodes)++;                                                                 /* the (K+1)-mer goes to KmerSetsPatch, :404-463 */
		(*edge_c)++;
		(*edge_counter)++;
		const sdto_kmer wordplus = sdto_kmer_plus(first->kmer, sdto_last_char(last->kmer));
		const sdto_kmer bal_wordplus = sdto_rc_kplus1(wordplus, K, S->nw);
		if (sdto_kmer_smaller(wordplus, bal_wordplus))
			patch_put(S, &wordplus, (uint32_t)*edge_c, bal_edge + 1);
		else
			patch_put(S, &bal_wordplus, (uint32_t)(*edge_c + bal_edge), -bal_edge + 1);
		symbol = first->node->count;                                                      /* :470-473 */
	} else {
		(*edge_c)++;
		(*edge_counter)++;
	}
	for (int i = n - 2; i >= 1; i--) {                                                    /* :488-513: the interior, last to first */
		sdto_node *d = b[i].node;
		symbol += length == 1 ? (long long)d->count : (long long)left_covs(d);            /* (reads a link word that may hold an edge id) */
		if (b[i].smaller) nd_set_edge(d, (uint32_t)*edge_c, bal_edge + 1);                 /* :497-507: inEdge, l_links = id, twin */
		else nd_set_edge(d, (uint32_t)(*edge_c + bal_edge), -bal_edge + 1);
		seq[ci--] = (char)sdto_last_char(b[i].kmer);
	}
	long long cvg;
	if (length > 1) cvg = symbol / (length - 1) * 10;
	else cvg = symbol / length * 10;
	if (cvg > 16000) cvg = 16000;                                                         /* MaxEdgeCov, inc/def.h:37 */
	/* output_1edge, output_pregraph.c:83-100 */
	fprintf(fp, ">length %d,", length);
	print_kmer_sep(fp, S, first->kmer, ',');
	print_kmer_sep(fp, S, last->kmer, ',');
	fprintf(fp, "cvg %d, %d\n", (int)cvg, bal_edge);
	for (int i = 0; i < length; i++) {
		fputc("ACTG"[(int)seq[i]], fp);
		if ((i + 1) % 100 == 0) fputc('\n', fp);
	}
	fputc('\n', fp);
	*edge_c += bal_edge;
}

/* startEdgeFromNode :193-310 for one side of one node; returns the number of beads */
static int string_beads(sdto_sets *S, sdto_node *n1, int right, int ch1, sdto_bead **beads, size_t *cap)
{
	const int K = S->K;
	size_t n = 0;
	sdto_bead *b = *beads;
	sdto_kmer oriented = right ? n1->seq : sdto_reverse_complement(n1->seq, K);
	b[n].node = n1; b[n].kmer = oriented; b[n].smaller = right; n++;
	sdto_kmer word = sdto_next_kmer(oriented, right ? ch1 : (ch1 ^ 2), K), canon;
	int smaller, ch;
	sdto_node *o = find_oriented(S, word, &smaller, &canon);
	while (o->linear) {
		if (n + 2 > *cap) { *cap *= 2; b = *beads = (sdto_bead *)realloc(b, *cap * sizeof *b); }
		oriented = smaller ? canon : sdto_reverse_complement(canon, K);
		b[n].node = o; b[n].kmer = oriented; b[n].smaller = smaller; n++;
		if (smaller) {
			for (ch = 0; ch < 4; ch++) if (rcov(o, ch)) break;
			word = sdto_next_kmer(oriented, ch, K);
		} else {
			for (ch = 0; ch < 4; ch++) if (lcov(o, ch)) break;
			word = sdto_next_kmer(oriented, ch ^ 2, K);
		}
		o = find_oriented(S, word, &smaller, &canon);
	}
	b[n].node = o; b[n].kmer = smaller ? canon : sdto_reverse_complement(canon, K); b[n].smaller = smaller; n++;
	return (int)n;
}

uint64_t sdto_write_edges(sdto_sets *S, const char *path, uint64_t *edge_counter_out, uint64_t *extra_nodes_out)
{
	FILE *fp = fopen(path, "w");
	if (!fp) return 0;
	const int K = S->K;
	long long edge_c = 0;
	uint64_t edge_counter = 0, extra = 0;
	size_t cap = 1024, seqcap = 1024;
	sdto_bead *beads = (sdto_bead *)malloc(cap * sizeof *beads);
	char *seqbuf = (char *)malloc(seqcap);
	for (int t = 0; t < S->nsets; t++) {                                                  /* make_edge :312-349 */
		sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			sdto_node *n1 = s->array + i;
			if (n1->linear || n1->deleted) continue;                                      /* :201-204 */
			for (int side = 0; side < 2; side++)                                          /* outgoing list, then incoming list */
				for (int ch1 = 0; ch1 < 4; ch1++) {
					if (!(side == 0 ? rcov(n1, ch1) : lcov(n1, ch1))) continue;            /* live: earlier edges have zeroed links */
					const int n = string_beads(S, n1, side == 0, ch1, &beads, &cap);
					int palindrome = 1;                                                   /* check_iden_kmerList :563-588 */
					for (int j = 0; j < n && palindrome; j++) {
						const sdto_kmer a = beads[n - 1 - j].kmer, bb = sdto_reverse_complement(beads[j].kmer, K);
						if (memcmp(a.w, bb.w, sizeof a.w) != 0) palindrome = 0;
					}
					merge_linear(S, fp, beads, n, palindrome ? 0 : 1, &edge_c, &edge_counter, &extra, &seqbuf, &seqcap);
				}
		}
	}
	fclose(fp);
	free(beads);
	free(seqbuf);
	if (edge_counter_out) *edge_counter_out = edge_counter;
	if (extra_nodes_out) *extra_nodes_out = extra;
	S->num_ed = (uint64_t)edge_c;
	return (uint64_t)edge_c;
}

/* ---- prlRead2edge (prlRead2path.c:817-1335; per read: chopKmer4read :253-371, searchKmer, parse1read :617-789,
 * search1kmerPlus :575-615, thread_add1preArc :413-428; output_arcs :454-505) after sdto_write_edges: every read -> its path
 * of edge ids -> one arc per adjacent pair; <prefix>.preArc lists, per from-edge in id order, its arcs with the one seen
 * first LAST (new arcs are pushed at the head of the list).  -n (N_kmer) is off, as everywhere in this build. */
typedef struct sdto_arc { uint32_t to, mult; struct sdto_arc *next; } sdto_arc;

uint64_t sdto_read2edge(sdto_sets *S, const uint8_t *codes, const uint64_t *offs, uint64_t nreads, const char *path)
{
	const int K = S->K;
	sdto_arc **heads = (sdto_arc **)calloc(S->num_ed + 2, sizeof *heads);
	uint64_t narcs = 0;
	size_t cap = 1024;
	uint64_t *mix = (uint64_t *)malloc(cap * sizeof *mix);
	sdto_kmer *plus = (sdto_kmer *)malloc(cap * sizeof *plus);
	uint8_t *flag = (uint8_t *)malloc(cap), *psm = (uint8_t *)malloc(cap);
	for (uint64_t t = 0; t < nreads; t++) {
		const uint8_t *seq = codes + offs[t];
		const int len = (int)(offs[t + 1] - offs[t]);
		if (len < K + 1) continue;                                                        /* :969 */
		const int n = len - K + 1;
		if ((size_t)n + 1 > cap) {
			cap = (size_t)n * 2;
			mix = (uint64_t *)realloc(mix, cap * sizeof *mix);
			plus = (sdto_kmer *)realloc(plus, cap * sizeof *plus);
			flag = (uint8_t *)realloc(flag, cap);
			psm = (uint8_t *)realloc(psm, cap);
		}
		/* parse1read :617-789 over the read's k-mers (chopped and looked up on the fly) */
		sdto_kmer word;
		memset(&word, 0, sizeof word);
		for (int i = 0; i < K - 1; i++) word = sdto_next_kmer(word, seq[i], K);
		unsigned retain = 0, pos = 0;
		int is_prev = 0;
		sdto_kmer prev_kmer;
		memset(&prev_kmer, 0, sizeof prev_kmer);
		for (int j = 0; j < n; j++) {
			word = sdto_next_kmer(word, seq[j + K - 1], K);
			int smaller;
			sdto_node *node = find_oriented(S, word, &smaller, NULL);
			if (node->deleted || (node->linear && !ND_INEDGE(node))) {                    /* deleted, or in a floating loop */
				if (retain < 2) { retain = 0; pos = 0; } else break;
				continue;                                                                 /* (is_prev is NOT reset: upstream) */
			}
			if (node->linear) {
				const uint64_t edge_index = smaller ? node->l_links : node->l_links + ND_TWIN(node) - 1;
				if (retain == 0 || is_prev) {
					retain++; mix[pos] = edge_index; flag[pos++] = 0; is_prev = 0;
				} else if (edge_index != mix[pos - 1]) {
					retain++; mix[pos] = edge_index; flag[pos++] = 0;
				}
			} else {
				const sdto_kmer cur = smaller ? node->seq : sdto_reverse_complement(node->seq, K);
				if (is_prev) {
					retain++;
					const sdto_kmer wp = sdto_kmer_plus(prev_kmer, sdto_last_char(cur)), bwp = sdto_rc_kplus1(wp, K, S->nw);
					if (sdto_kmer_smaller(wp, bwp)) { psm[pos] = 1; plus[pos] = wp; }
					else { psm[pos] = 0; plus[pos] = bwp; }
					mix[pos] = 1;                                                         /* (non-zero: resolved below) */
					flag[pos++] = 1;
				}
				is_prev = 1;
				prev_kmer = cur;
			}
		}
		if (retain < 2) continue;                                                         /* :771-776: no path */
		if (pos < (unsigned)n) { flag[pos] = 0; mix[pos] = 0; }
		const unsigned end = pos < (unsigned)n ? pos + 1 : pos;
		/* search1kmerPlus :575-615 for the (K+1)-mers, up to the terminator */
		for (unsigned j = 0; j < end; j++) {
			if (!flag[j]) { if (mix[j] == 0) break; continue; }
			const sdto_patch *pn = patch_find(S, &plus[j]);
			mix[j] = pn ? (psm[j] ? pn->edge : pn->edge + pn->twin - 1) : 0;
		}
		/* arcs :190-241 */
		for (unsigned j = 0; j + 1 < end; j++) {
			if (mix[j] == 0 || mix[j + 1] == 0) break;
			const uint32_t from = (uint32_t)mix[j], to = (uint32_t)mix[j + 1];
			sdto_arc *a = heads[from];
			while (a && a->to != to) a = a->next;
			if (a) {
				a->mult++;
			} else {
				a = (sdto_arc *)malloc(sizeof *a);
				a->to = to; a->mult = 1; a->next = heads[from];                           /* prlAllocatePreArc: multiplicity 1, pushed at the head */
				heads[from] = a;
				narcs++;
			}
		}
	}
	FILE *fp = fopen(path, "w");                                                         /* output_arcs :454-505 */
	if (fp) {
		for (uint64_t i = 1; i <= S->num_ed; i++) {
			if (!heads[i]) continue;
			fprintf(fp, "%u", (unsigned)i);
			for (sdto_arc *a = heads[i]; a; a = a->next) fprintf(fp, " %u %u", a->to, a->mult);
			fputc('\n', fp);
		}
		fclose(fp);
	}
	for (uint64_t i = 0; i <= S->num_ed + 1; i++)
		for (sdto_arc *a = heads[i]; a;) { sdto_arc *nx = a->next; free(a); a = nx; }
	free(heads); free(mix); free(plus); free(flag); free(psm);
	return narcs;
}

uint64_t sdto_write_vertex(const sdto_sets *S, const char *path)
{
	FILE *fp = fopen(path, "w");
	if (!fp) return 0;
	uint64_t n = 0;
	for (int t = 0; t < S->nsets; t++) {
		const sdto_set *s = S->sets[t];
		for (uint64_t i = 0; i < s->size; i++) {
			if (FL_NULL(s->flags, i)) continue;
			const sdto_node *rs = s->array + i;
			if (rs->linear || rs->deleted) continue;
			n++;
			const uint64_t *w = rs->seq.w;
			if (S->nw == 4)
				fprintf(fp, "%llx %llx %llx %llx", (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3]);
			else if (S->nw == 2)
				fprintf(fp, "%llx %llx", (unsigned long long)w[2], (unsigned long long)w[3]);
			else if (w[3])
				fprintf(fp, "%llx", (unsigned long long)w[3]);
			else
				fprintf(fp, "0x0");
			fputc(' ', fp);
			if (n % 8 == 0) fputc('\n', fp);
		}
	}
	fputc('\n', fp);
	fclose(fp);
	return n;
}
