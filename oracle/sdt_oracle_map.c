/*
 * sdt_oracle_map.c -- CPU restatement of the `map` stage's hashing path: prlContig2nodes (contig k-mers ->
 * KmerSets with a contig id / position payload) and prlRead2Ctg (reads -> k-mer look-ups -> contig hits ->
 * *.readOnContig / *.ctg2Read / *.readInGap / *.peGrads [/ *.readInformation]).
 *
 * TEST INFRASTRUCTURE ONLY (see sdt_oracle.h).  Pinned by tests/test_oracle_vs_reference.py against the files the
 * reference's own `map` wrote for the cases under tests/golden/map_cases (tests/golden/make_map_golden.py).
 * Citations are relative to /root/reference/src.
 */
#include "sdt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAP_MAX_HITS 20                       /* pos_temp[20] / alignLen[20], prlRead2Ctg.c:240-241 */

struct sdto_map_s {
	sdto_sets *S;                              /* node.l_links = contig id, r_links = pos (24 bit), pad = twin */
	uint32_t *ctg_len;                         /* contig_array[i].length, i = 1..num_ctg (basicContigInfo :610-648) */
	int32_t *ctg_bal;                          /* contig_array[i].bal_edge */
	uint32_t num_ctg;
	uint64_t kmers;
};

sdto_map *sdto_map_new(int nsets, int nw, int K)
{
	sdto_map *M = (sdto_map *)calloc(1, sizeof *M);
	M->S = sdto_sets_new(nsets, nw, K);
	return M;
}

void sdto_map_free(sdto_map *M)
{
	if (!M) return;
	sdto_sets_free(M->S);
	free(M->ctg_len);
	free(M->ctg_bal);
	free(M);
}

/* basicContigInfo (prlRead2Ctg.c:610-648): one line "index length bal" per contig or contig pair of *.ContigIndex */
void sdto_map_set_contig_index(sdto_map *M, const uint32_t *index_length, const int32_t *index_bal, uint32_t nlines, uint32_t num_all)
{
	M->num_ctg = num_all;
	M->ctg_len = (uint32_t *)calloc((size_t)num_all + 2, sizeof(uint32_t));
	M->ctg_bal = (int32_t *)calloc((size_t)num_all + 2, sizeof(int32_t));
	uint32_t k = 0;
	for (uint32_t i = 0; i < nlines && k < num_all; i++) {
		M->ctg_len[++k] = index_length[i];
		M->ctg_bal[k] = index_bal[i] + 1;
		if (index_bal[i] == 0) continue;
		if (k >= num_all) break;
		M->ctg_len[++k] = index_length[i];
		M->ctg_bal[k] = -index_bal[i] + 1;
	}
}

static uint32_t twin_ctg(const sdto_map *M, uint32_t c) { return (uint32_t)((int64_t)c + M->ctg_bal[c] - 1); }   /* attachPEinfo.c:479 */

/* chopKmer4read + singleKmer for one contig (prlHashCtg.c:175-262, 110-139): the FIRST occurrence of a k-mer
 * sets contig id / position / strand; every later occurrence only marks the node deleted */
void sdto_map_add_contig(sdto_map *M, const uint8_t *codes, int len, uint32_t ctg_id)
{
	sdto_sets *S = M->S;
	const int K = S->K, nw = S->nw;
	if (len < K) return;
	const int n = len - K + 1;
	sdto_kmer *keys = (sdto_kmer *)malloc((size_t)n * sizeof(sdto_kmer));
	uint8_t *pc = (uint8_t *)malloc((size_t)n), *nc = (uint8_t *)malloc((size_t)n), *smaller = (uint8_t *)malloc((size_t)n);
	uint64_t *hash = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
	/* the same rolling chop as pass 1; `smaller` restated here: word < its reverse complement (:205-216) */
	sdto_kmer word = {{0, 0, 0, 0}};
	for (int i = 0; i < K; i++) word = sdto_next_kmer(word, codes[i], K);
	for (int j = 0; j < n; j++) {
		if (j) word = sdto_next_kmer(word, codes[j - 1 + K], K);
		const sdto_kmer bal = sdto_reverse_complement(word, K);
		smaller[j] = (uint8_t)sdto_kmer_smaller(word, bal);
		keys[j] = smaller[j] ? word : bal;
		hash[j] = sdto_hash_kmer(keys[j], nw);
	}
	for (int j = 0; j < n; j++) {
		sdto_set *set = S->sets[hash[j] % (uint64_t)S->nsets];
		uint64_t slot;
		const int existed = sdto_set_put(set, keys[j], 4, 4, nw, &slot);
		sdto_node *node = &set->array[slot];
		if (!existed) {
			node->pad = smaller[j] ? 0 : 1;                      /* twin */
			node->l_links = ctg_id;
			node->r_links = (uint32_t)j & 0xFFFFFFu;             /* 24-bit bitfield */
		} else {
			node->deleted = 1;
		}
	}
	M->kmers += (uint64_t)n;
	free(keys); free(pc); free(nc); free(smaller); free(hash);
}

void sdto_map_index_counts(const sdto_map *M, uint64_t *nodes, uint64_t *kmers)
{
	*nodes = sdto_sets_node_count(M->S);
	*kmers = M->kmers;
}

/* chopKmer4read + searchKmer + parse1read for one read (prlRead2Ctg.c:129-212, 214-229, 231-353).
 * Returns the number of recorded hits (0 = not mapped), or -1 when the read has more than 20 candidate contigs
 * (the reference then writes past pos_temp[20]: undefined). */
int sdto_map_read(const sdto_map *M, const uint8_t *codes, int len, int align_len, sdto_hit hits[MAP_MAX_HITS], int *best,
                  int *footprint)
{
	const sdto_sets *S = M->S;
	const int K = S->K, nw = S->nw;
	*footprint = 0;
	*best = -1;
	if (len < K + 1) return 0;
	const int n = len - K + 1;
	const sdto_node **node = (const sdto_node **)calloc((size_t)n, sizeof(*node));
	uint8_t *smaller = (uint8_t *)malloc((size_t)n);
	sdto_kmer word = {{0, 0, 0, 0}};
	for (int i = 0; i < K; i++) word = sdto_next_kmer(word, codes[i], K);
	for (int j = 0; j < n; j++) {
		if (j) word = sdto_next_kmer(word, codes[j - 1 + K], K);
		const sdto_kmer bal = sdto_reverse_complement(word, K);
		smaller[j] = (uint8_t)sdto_kmer_smaller(word, bal);
		const sdto_kmer key = smaller[j] ? word : bal;
		const sdto_set *set = S->sets[sdto_hash_kmer(key, nw) % (uint64_t)S->nsets];
		uint64_t slot;
		if (sdto_set_search(set, key, nw, &slot) && !set->array[slot].deleted) node[j] = &set->array[slot];
	}
	const int alldgn = len > align_len ? align_len : len;
	const int multi = alldgn - K + 1 < 5 ? 5 : alldgn - K + 1;
	int counter = 0, counter2 = 0, max_occ = 0, nh = 0, overflow = 0;
	for (int j = 0; j < n; j++) {
		if (!node[j]) continue;
		int flag = 1;
		for (int s = j + 1; s < n; s++)
			if (node[s] && node[s]->l_links == node[j]->l_links) { flag++; node[s] = NULL; }
		if (flag >= 2) counter2++;
		if (flag < multi) continue;
		counter++;
		if (nh >= MAP_MAX_HITS) { overflow = 1; break; }
		const uint32_t ctg = node[j]->l_links, pos = node[j]->r_links;
		sdto_hit *h = &hits[nh];
		h->readOffset = (uint32_t)j + 1;
		h->alignLength = (uint32_t)flag;
		if ((int)node[j]->pad == (int)smaller[j]) {
			h->orien = '-';
			h->contigID = twin_ctg(M, ctg);
			h->contigOffset = (int32_t)(M->ctg_len[ctg] - pos - (uint32_t)K);
		} else {
			h->orien = '+';
			h->contigID = ctg;
			h->contigOffset = (int32_t)pos;
		}
		if (flag > max_occ) { max_occ = flag; *best = nh; }
		nh++;
	}
	free(node);
	free(smaller);
	if (overflow) return -1;
	if (!counter) return 0;
	if (counter2 > 1) *footprint = 1;
	return nh;
}

/* writeChar2tightString (seq.c:49-71) */
static void tight_put(char nt, char *tight, int pos)
{
	char *byte = tight + pos / 4;
	switch (pos % 4) {
	case 0: *byte &= 63; *byte += nt << 6; return;
	case 1: *byte &= 207; *byte += nt << 4; return;
	case 2: *byte &= 243; *byte += nt << 2; return;
	default: *byte &= 252; *byte += nt; return;
	}
}

typedef struct {
	FILE *gap;
	char *rc1;                  /* rcSeq[1]: thread 0's reverse-complement scratch AND the tight-string buffer (:430-437) */
	long long reads_in_gap;
	FILE *fill_gap, *fill_pe;   /* -f: the CONTENT of shortreadInGap.gz / PEreadOnContig.gz, uncompressed */
} gap_out;

/* output1read (prlRead2Ctg.c:423-446) */
static void output1read(gap_out *G, const uint8_t *codes, int len, int ctg, int pos, char orien, int ins, int dhflag)
{
	G->reads_in_gap++;
	for (int i = 0; i < len; i++) tight_put((char)codes[i], G->rc1, i);
	fwrite(&len, sizeof(int), 1, G->gap);
	fwrite(&ctg, sizeof(int), 1, G->gap);
	fwrite(&pos, sizeof(int), 1, G->gap);
	fwrite(G->rc1, 1, (size_t)(len / 4 + 1), G->gap);
	if (G->fill_gap && ins < 2000 && len > 0) {                                    /* :439-444 */
		fprintf(G->fill_gap, ">%d\t%d\t%d\t%c\t%d\t%d\n", len, ctg, pos, orien, ins, dhflag);
		for (int i = 0; i < len; i++) fputc("ACTG"[codes[i]], G->fill_gap);
		fputc('\n', G->fill_gap);
	}
}

/* getPEreadOnContig (:493-524): both mates mapped; the tight strings go through the same shared buffer */
static void pe_on_contig(gap_out *G, const uint8_t *c1, int len1, int ctg1, int pos1, char o1, int ins1,
                         const uint8_t *c2, int len2, int ctg2, int pos2, char o2, int ins2)
{
	if (!(ins2 < 2000 && ins2 == ins1)) return;
	fwrite(&len1, sizeof(int), 1, G->fill_pe);
	fwrite(&ctg1, sizeof(int), 1, G->fill_pe);
	fwrite(&pos1, sizeof(int), 1, G->fill_pe);
	fwrite(&o1, 1, 1, G->fill_pe);
	fwrite(&ins1, sizeof(int), 1, G->fill_pe);
	for (int i = 0; i < len1; i++) tight_put((char)c1[i], G->rc1, i);
	fwrite(G->rc1, 1, (size_t)(len1 / 4 + 1), G->fill_pe);
	fwrite(&len2, sizeof(int), 1, G->fill_pe);
	fwrite(&ctg2, sizeof(int), 1, G->fill_pe);
	fwrite(&pos2, sizeof(int), 1, G->fill_pe);
	fwrite(&o2, 1, 1, G->fill_pe);
	fwrite(&ins2, sizeof(int), 1, G->fill_pe);
	for (int i = 0; i < len2; i++) tight_put((char)c2[i], G->rc1, i);
	fwrite(G->rc1, 1, (size_t)(len2 / 4 + 1), G->fill_pe);
}

/* prlRead2Ctg's main loop + recordAlldgn (prlRead2Ctg.c:561-608 part of it, 656-860) over reads that are already
 * coded, in consumption order (read1, read2, read1, ...).  lib_of_read[i] indexes lib_ins / lib_map_len.
 * buffer_size: the reference's 100000000 (k-mers per batch); smaller values exercise the batch logic.
 * counters[0] = reads, [1] = mapped, [2] = reads in gap, [3] = reads with > 20 candidate contigs (undefined upstream). */
int sdto_map_run(const sdto_map *M, const uint8_t *codes, const uint64_t *offsets, uint64_t nreads, const int32_t *lib_of_read,
                 const int32_t *lib_ins, const int32_t *lib_map_len, int max_read_len, int thrd_num, int buffer_size,
                 int read_trace, int fill, const char *prefix, long long counters[4])
{
	const int K = M->S->K;
	char name[4200];
	snprintf(name, sizeof name, "%s.readInGap", prefix);
	gap_out G = {fopen(name, "wb"), (char *)calloc((size_t)max_read_len + 8, 1), 0, NULL, NULL};
	if (fill) {
		snprintf(name, sizeof name, "%s.shortreadInGap", prefix);
		G.fill_gap = fopen(name, "w");
		snprintf(name, sizeof name, "%s.PEreadOnContig", prefix);
		G.fill_pe = fopen(name, "wb");
		if (!G.fill_gap || !G.fill_pe) return -1;
	}
	snprintf(name, sizeof name, "%s.readOnContig", prefix);
	FILE *fo = fopen(name, "w");
	snprintf(name, sizeof name, "%s.ctg2Read", prefix);
	FILE *f3 = fopen(name, "w");
	FILE *f4 = NULL;
	if (read_trace) { snprintf(name, sizeof name, "%s.readInformation", prefix); f4 = fopen(name, "w"); }
	if (!G.gap || !fo || !f3 || (read_trace && !f4)) return -1;
	fprintf(fo, "read\tcontig\tpos\n");
	fprintf(f3, "read\tcontig\tpos\n");
	int max_read_num = buffer_size / (max_read_len - K + 1);
	if (max_read_num % 2) max_read_num--;
	long long read_counter = 0, map_counter = 0, overflowed = 0;
	/* per-batch state */
	int *ctg_id = (int *)calloc((size_t)max_read_num + 1, sizeof(int)), *posv = (int *)calloc((size_t)max_read_num + 1, sizeof(int));
	int *nh = (int *)calloc((size_t)max_read_num + 1, sizeof(int)), *foot = (int *)calloc((size_t)max_read_num + 1, sizeof(int));
	char *orien = (char *)calloc((size_t)max_read_num + 1, 1);     /* orienArray: written for mapped reads only, so an unmapped read shows
	                                                                 * what an earlier batch left at its index (:318-327) */
	sdto_hit *hits = (sdto_hit *)calloc(((size_t)max_read_num + 1) * MAP_MAX_HITS, sizeof(sdto_hit));
	int align_len = 0, prev_lib = -1;
	uint64_t start = 0;
	while (start < nreads) {
		/* the main thread reads up to maxReadNum reads; ALIGNLEN is a global that keeps changing while it reads and
		 * parse1read sees the value left by the LAST read of the batch (:774-791) */
		uint64_t end = start + (uint64_t)max_read_num;
		if (end > nreads) end = nreads;
		for (uint64_t r = start; r < end; r++) {
			const int lib = lib_of_read[r], len = (int)(offsets[r + 1] - offsets[r]);
			const int ins = lib_ins[lib];
			if (lib != prev_lib) {
				prev_lib = lib;
				align_len = lib_map_len[lib];
				if (ins > 1000) align_len = align_len < 35 ? 35 : align_len;
				else align_len = align_len < 32 ? 32 : align_len;
			}
			if (ins > 1000) align_len = align_len < (len / 2 + 1) ? (len / 2 + 1) : align_len;
		}
		const int rc = (int)(end - start);
		/* signal 2: thread 0 chops reads 0, thrd_num, 2*thrd_num, ... and leaves their reverse complement in rcSeq[1] */
		for (int t = 0; t < rc; t += thrd_num) {
			const uint8_t *s = codes + offsets[start + (uint64_t)t];
			const int len = (int)(offsets[start + (uint64_t)t + 1] - offsets[start + (uint64_t)t]);
			if (len < K + 1) continue;
			for (int i = 0; i < len; i++) G.rc1[i] = (char)(s[len - 1 - i] ^ 2);            /* reverseComplementSeq seq.c:93-109 */
		}
		/* signals 1 + 3 */
		for (int t = 0; t < rc; t++) {
			const uint8_t *s = codes + offsets[start + (uint64_t)t];
			const int len = (int)(offsets[start + (uint64_t)t + 1] - offsets[start + (uint64_t)t]);
			int best;
			nh[t] = sdto_map_read(M, s, len, align_len, hits + (size_t)t * MAP_MAX_HITS, &best, &foot[t]);
			if (nh[t] < 0) { overflowed++; nh[t] = 0; foot[t] = 0; }
			if (nh[t] > 0) {
				const sdto_hit *h = &hits[(size_t)t * MAP_MAX_HITS + best];
				ctg_id[t] = (int)h->contigID;
				posv[t] = h->contigOffset - (int)h->readOffset + 1;
				orien[t] = h->orien;
			} else {
				ctg_id[t] = 0;
			}
		}
		/* recordAlldgn (:526-608) */
		for (int t = 0; t < rc; t++) {
			const uint64_t r = start + (uint64_t)t;
			read_counter++;
			int rd1gap = 0, rd2gap = 0;
			const int ctg = ctg_id[t];
			if (t % 2 == 1) {
				if (ctg_id[t] < 1 && ctg_id[t - 1] > 0) {                        /* read 2 in gap: getReadIngap(t, .., 0) :448-482 */
					const int len2 = (int)(offsets[r + 1] - offsets[r]);
					ctg_id[t] = ctg_id[t - 1];
					posv[t] = posv[t - 1] + lib_ins[lib_of_read[r]] - len2;
					output1read(&G, codes + offsets[r], len2, ctg_id[t], posv[t], orien[t - 1] == '+' ? '-' : '+', lib_ins[lib_of_read[r]], 1);
					rd2gap = 1;
				} else if (ctg_id[t] > 0 && ctg_id[t - 1] < 1) {                 /* read 1 in gap */
					const int len1 = (int)(offsets[r] - offsets[r - 1]);
					ctg_id[t - 1] = ctg_id[t];
					posv[t - 1] = posv[t] + lib_ins[lib_of_read[r - 1]] - len1;
					output1read(&G, codes + offsets[r - 1], len1, ctg_id[t - 1], posv[t - 1], orien[t] == '+' ? '-' : '+', lib_ins[lib_of_read[r - 1]], 1);
					rd1gap = 1;
				} else if (ctg_id[t] > 0 && ctg_id[t - 1] > 0 && fill) {          /* PE read on contig :554-558 */
					pe_on_contig(&G, codes + offsets[r - 1], (int)(offsets[r] - offsets[r - 1]), ctg_id[t - 1], posv[t - 1], orien[t - 1], lib_ins[lib_of_read[r - 1]],
					             codes + offsets[r], (int)(offsets[r + 1] - offsets[r]), ctg_id[t], posv[t], orien[t], lib_ins[lib_of_read[r]]);
				}
			}
			if (ctg < 1) continue;
			map_counter++;
			const sdto_hit *H = &hits[(size_t)t * MAP_MAX_HITS];
			const sdto_hit *h = (read_counter % 2 == 1) ? &H[nh[t] - 1] : &H[0];
			fprintf(fo, "%lld\t%u\t%d\t%c\n", read_counter, h->contigID, h->contigOffset - (int)h->readOffset + 1, h->orien);
			for (int m = 0; m < nh[t]; m++) {
				if (H[m].alignLength >= 5)
					fprintf(f3, "%lld\t%u\t%d\t%c\n", read_counter, H[m].contigID, (int)H[m].readOffset - H[m].contigOffset, H[m].orien);
				if (read_trace && H[m].alignLength >= 5) {
					const int span = (int)H[m].alignLength + K - 1;
					if (H[m].orien == '+')
						fprintf(f4, "%lld\t%d\t%llu\t%d\t%d\t%c\n", read_counter, (int)H[m].readOffset - 1, (unsigned long long)H[m].contigID,
						        H[m].contigOffset, span, H[m].orien);
					else
						fprintf(f4, "%lld\t%d\t%llu\t%d\t%d\t%c\n", read_counter, (int)H[m].readOffset - 1,
						        (unsigned long long)twin_ctg(M, H[m].contigID), (int)M->ctg_len[H[m].contigID] - H[m].contigOffset - span, span, H[m].orien);
				}
			}
			if (t % 2 == 0) continue;
			/* "reads are not located by pe info but across edges" (:591-606); locate1read is unreachable: a footprint read is mapped */
			if (foot[t - 1] && !rd1gap)
				output1read(&G, codes + offsets[r - 1], (int)(offsets[r] - offsets[r - 1]), ctg_id[t - 1], posv[t - 1], orien[t] == '+' ? '-' : '+',
				            lib_ins[lib_of_read[r - 1]], 1);
			if (foot[t] && !rd2gap)
				output1read(&G, codes + offsets[r], (int)(offsets[r + 1] - offsets[r]), ctg_id[t], posv[t], orien[t - 1] == '+' ? '-' : '+',
				            lib_ins[lib_of_read[r]], 2);
		}
		start = end;
	}
	counters[0] = read_counter;
	counters[1] = map_counter;
	counters[2] = G.reads_in_gap;
	counters[3] = overflowed;
	fclose(G.gap); fclose(fo); fclose(f3);
	if (f4) fclose(f4);
	if (G.fill_gap) fclose(G.fill_gap);
	if (G.fill_pe) fclose(G.fill_pe);
	free(G.rc1); free(ctg_id); free(posv); free(nh); free(foot); free(hits); free(orien);
	return 0;
}
