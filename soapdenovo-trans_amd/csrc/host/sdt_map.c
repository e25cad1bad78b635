/* sdt_map.c -- the reference's `map` command line over libsdt_gpu.so.
 *
 *   sdt-map [map] -s configFile -g inputGraph [-p n_cpu] [-K kmer] [-r] [--max-k 31|63|127] [--device n]
 *
 * Mirrors call_align (map.c:62-103): getMinOverlap (<g>.preGraphBasic) -> prlContig2nodes (<g>.contig -> k-mer
 * index with contig id / position, prlHashCtg.c:287-425) -> prlRead2Ctg (paired reads of the asm_flags 2|3
 * libraries -> <g>.readOnContig, <g>.ctg2Read, <g>.readInGap, <g>.peGrads [, <g>.readInformation with -r],
 * prlRead2Ctg.c:656-894).  The hashing and the per-read alignment (chop, look-up, parse1read) run on the GPU
 * (sdt_gpu_index_contigs / sdt_gpu_align_reads); this file keeps the reference's read order, batch geometry
 * (ALIGNLEN is a global that parse1read samples per 10^8-k-mer batch), stdout lines and file writers.
 * -f adds <g>.shortreadInGap.gz and <g>.PEreadOnContig.gz (:439-444, 493-524).
 * Not supported: b= BAM input; single-end files are ignored as in the reference's map.
 * Written from the reference's behaviour; no reference code is used. */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include "../sdt_knobs.h"
#include "../../../include/sdt_gpu.h"
#include "libcfg.h"
#include "seqio.h"
#include "graph/par.h"

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double g_t_last;
static void phase(const char *name)
{
	if (!sdt_env("SDT_TIMING")) return;
	const double t = now_ms();
	fprintf(stderr, "[sdt-map] %-32s %9.1f ms\n", name, t - g_t_last);
	g_t_last = t;
}

/* ---- reads kept on the host as the parser delivers them: one entry per parsed chunk ---- */
typedef struct {
	uint32_t *words;
	uint64_t *offs;
	uint64_t nreads, nwords;
	int32_t *alen;            /* ALIGNLEN of each read's batch */
	uint64_t *info;           /* sdt_gpu_align_reads read_info */
	sdt_hit *hits;
	uint64_t nhits;
} mbatch;

typedef struct {
	mbatch *b;
	size_t n, cap;
	uint64_t nreads;
	uint64_t *first;          /* first[i] = index of the first read of batch i (n + 1 entries), built after loading */
} mfile;

typedef struct {
	int lib, paired;          /* paired: A and B alternate; else A alone (interleaved file) */
	mfile A, B;
	uint64_t base;            /* global index of its first read */
	uint64_t nreads;
} mstream;

static int keep_batch(void *user, const sdt_batch *b)
{
	mfile *f = (mfile *)user;
	if (f->n == f->cap) {
		f->cap = f->cap ? f->cap * 2 : 64;
		f->b = (mbatch *)realloc(f->b, f->cap * sizeof(mbatch));
	}
	mbatch *m = &f->b[f->n++];
	memset(m, 0, sizeof *m);
	m->nreads = b->nreads;
	m->nwords = b->nwords;
	m->words = (uint32_t *)malloc(b->nwords * sizeof(uint32_t));
	memcpy(m->words, b->words, b->nwords * sizeof(uint32_t));
	m->offs = (uint64_t *)malloc((b->nreads + 1) * sizeof(uint64_t));
	memcpy(m->offs, b->offsets, (b->nreads + 1) * sizeof(uint64_t));
	f->nreads += b->nreads;
	return 0;
}

static void index_file(mfile *f)
{
	f->first = (uint64_t *)malloc((f->n + 1) * sizeof(uint64_t));
	f->first[0] = 0;
	for (size_t i = 0; i < f->n; i++) f->first[i + 1] = f->first[i] + f->b[i].nreads;
}

typedef struct { mbatch *b; uint64_t i; } rref;          /* one read: batch + index inside it */

static rref file_read(const mfile *f, uint64_t r)
{
	size_t lo = 0, hi = f->n;                              /* largest batch with first <= r */
	while (hi - lo > 1) {
		const size_t mid = (lo + hi) >> 1;
		if (f->first[mid] <= r) lo = mid; else hi = mid;
	}
	rref x = {&f->b[lo], r - f->first[lo]};
	return x;
}

static inline int rlen(rref x) { return (int)(x.b->offs[x.i + 1] - x.b->offs[x.i]); }
static inline unsigned rbase(rref x, int k)
{
	const uint64_t p = x.b->offs[x.i] + (uint64_t)k;
	return (x.b->words[p >> 4] >> (30 - 2 * (p & 15))) & 3u;
}

/* sequential walk over all reads in the order read1seqInLib hands them out */
typedef struct {
	const mstream *s;
	int ns, si;
	uint64_t k;               /* index inside the stream */
	size_t ba, bb;            /* current batch of A / B */
	uint64_t ia, ib;          /* next read inside those batches */
} cursor;

static int cur_next(cursor *c, rref *out, int *lib)
{
	while (c->si < c->ns && c->k >= c->s[c->si].nreads) { c->si++; c->k = 0; c->ba = c->bb = 0; c->ia = c->ib = 0; }
	if (c->si >= c->ns) return 0;
	const mstream *s = &c->s[c->si];
	const int useB = s->paired && (c->k & 1);
	const mfile *f = useB ? &s->B : &s->A;
	size_t *bi = useB ? &c->bb : &c->ba;
	uint64_t *ii = useB ? &c->ib : &c->ia;
	while (*ii >= f->b[*bi].nreads) { (*bi)++; *ii = 0; }
	out->b = &f->b[*bi];
	out->i = (*ii)++;
	*lib = s->lib;
	c->k++;
	return 1;
}

/* global read index -> read (used off the hot path only) */
static rref global_read(const mstream *S, int ns, uint64_t g)
{
	int si = 0;
	while (si + 1 < ns && S[si + 1].base <= g) si++;
	const mstream *s = &S[si];
	const uint64_t k = g - s->base;
	if (!s->paired) return file_read(&s->A, k);
	return file_read((k & 1) ? &s->B : &s->A, k >> 1);
}

/* ---- <g>.contig: FASTA, names "<id> length ..."; base coding as for reads (readseq1by1.c:47-120) ---- */
typedef struct {
	uint32_t *words;
	uint64_t nwords, nbases, cap;
	uint64_t *offs;
	uint32_t *ids;
	uint64_t n, ncap;
	long long num_seq;
	int max_len, min_len, name_len;
} contigs_t;

static void ctg_put(contigs_t *C, unsigned code)
{
	if ((C->nbases >> 4) + 8 >= C->cap) {
		const uint64_t ncap = C->cap ? C->cap * 2 : 1 << 16;
		C->words = (uint32_t *)realloc(C->words, ncap * sizeof(uint32_t));
		memset(C->words + C->cap, 0, (ncap - C->cap) * sizeof(uint32_t));
		C->cap = ncap;
	}
	C->words[C->nbases >> 4] |= (uint32_t)code << (30 - 2 * (C->nbases & 15));
	C->nbases++;
}

static int load_contigs(const char *path, int K, int len_cut, contigs_t *C)
{
	FILE *fp = fopen(path, "r");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", path); return -1; }
	memset(C, 0, sizeof *C);
	C->max_len = C->name_len = 10;                          /* prlHashCtg.c:303-304 */
	C->min_len = 1000;
	C->ncap = 1024;
	C->offs = (uint64_t *)malloc((C->ncap + 1) * sizeof(uint64_t));
	C->ids = (uint32_t *)malloc(C->ncap * sizeof(uint32_t));
	C->offs[0] = 0;
	char *line = NULL;
	size_t lcap = 0;
	ssize_t got;
	long long ordinal = 0;
	int have = 0, raw = 0;
	uint32_t id = 0;
	uint64_t rec_start = 0;
#define CLOSE_RECORD() do { \
		if (have) { \
			if (raw > C->max_len) C->max_len = raw; \
			if (raw < C->min_len) C->min_len = raw; \
			const uint64_t len = C->nbases - rec_start; \
			if (len < (uint64_t)K + 1 || len < (uint64_t)len_cut) { \
				/* dropped (:343-350): take its bases out of the stream again */ \
				for (uint64_t q = rec_start; q < C->nbases; q++) C->words[q >> 4] &= ~(3u << (30 - 2 * (q & 15))); \
				C->nbases = rec_start; \
			} else { \
				if (C->n == C->ncap) { C->ncap *= 2; C->offs = (uint64_t *)realloc(C->offs, (C->ncap + 1) * sizeof(uint64_t)); C->ids = (uint32_t *)realloc(C->ids, C->ncap * sizeof(uint32_t)); } \
				C->ids[C->n] = id > 0 ? id : (uint32_t)ordinal; \
				C->offs[++C->n] = C->nbases; \
			} \
		} } while (0)
	while ((got = getline(&line, &lcap, fp)) > 0) {
		if (line[0] == '#') continue;
		if (line[0] == '>') {
			CLOSE_RECORD();
			have = 1;
			ordinal++;
			raw = 0;
			rec_start = C->nbases;
			char name[512] = "";
			sscanf(line + 1, "%500s", name);
			const int nl = (int)strlen(name);
			if (nl > C->name_len) C->name_len = nl;
			id = (name[0] >= '0' && name[0] <= '9') ? (uint32_t)atoi(name) : 0;     /* getID :276-285 */
			continue;
		}
		if (!have) continue;
		raw += (int)strlen(line) - 1;                                             /* readseqpar :263 */
		for (ssize_t i = 0; i < got; i++) {
			unsigned char c = (unsigned char)line[i];
			if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 'a' + 'A');
			if (c >= 'A' && c <= 'Z') ctg_put(C, (unsigned)((c & 6) >> 1));
			else if (c == '.') ctg_put(C, 0);
		}
	}
	CLOSE_RECORD();
#undef CLOSE_RECORD
	C->num_seq = ordinal;
	free(line);
	fclose(fp);
	if (!C->words) { C->cap = 16; C->words = (uint32_t *)calloc(C->cap, sizeof(uint32_t)); }
	C->nwords = ((C->nbases + 15) >> 4) + 4;
	return 0;
}


static void cur_seek(cursor *c, const mstream *S, int ns, uint64_t g)
{
	memset(c, 0, sizeof *c);
	c->s = S;
	c->ns = ns;
	int si = 0;
	while (si + 1 < ns && S[si + 1].base <= g) si++;
	c->si = si;
	if (si >= ns || g >= S[si].base + S[si].nreads) { c->si = ns; return; }
	const mstream *s = &S[si];
	c->k = g - s->base;
	const uint64_t na = s->paired ? (c->k + 1) / 2 : c->k, nb = s->paired ? c->k / 2 : 0;
	rref a = file_read(&s->A, na);
	c->ba = (size_t)(a.b - s->A.b);
	c->ia = a.i;
	if (s->paired) {
		rref b = file_read(&s->B, nb);
		c->bb = (size_t)(b.b - s->B.b);
		c->ib = b.i;
	}
}

/* ---- text formatting of *.readOnContig / *.ctg2Read / *.readInformation, in parallel per block of reads ---- */
typedef struct { char *p; size_t n, cap; } tbuf;
static inline void tb_room(tbuf *b, size_t more)
{
	if (b->n + more <= b->cap) return;
	b->cap = b->cap ? b->cap * 2 : 1 << 20;
	while (b->cap < b->n + more) b->cap *= 2;
	b->p = (char *)realloc(b->p, b->cap);
}
static inline void tb_u64(tbuf *b, unsigned long long v)
{
	char tmp[24];
	int n = 0;
	do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	while (n) b->p[b->n++] = tmp[--n];
}
static inline void tb_i64(tbuf *b, long long v)
{
	if (v < 0) { b->p[b->n++] = '-'; tb_u64(b, (unsigned long long)(-v)); }
	else tb_u64(b, (unsigned long long)v);
}

typedef struct {
	const mstream *S;
	int ns, K, read_trace;
	const uint32_t *ctg_len, *ctg_twin;
	uint64_t total, block, wave_first;      /* reads per block; first block of this wave */
	tbuf *ro, *c2, *ri;                     /* one per block of the wave */
	long long *mapped, *overflowed;
} fmt_ctx;

static void fmt_blocks(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	fmt_ctx *F = (fmt_ctx *)vc;
	for (uint64_t bi = lo; bi < hi; bi++) {
		const uint64_t g0 = (F->wave_first + bi) * F->block, g1 = g0 + F->block < F->total ? g0 + F->block : F->total;
		/* work on private copies: the tbuf structs of neighbouring blocks share cache lines, and `n` moves per byte */
		tbuf ro_l = F->ro[bi], c2_l = F->c2[bi], ri_l = F->ri[bi];
		tbuf *ro = &ro_l, *c2 = &c2_l, *ri = &ri_l;
		ro->n = c2->n = ri->n = 0;
		long long mapped = 0, over = 0;
		cursor cu;
		cur_seek(&cu, F->S, F->ns, g0);
		rref x;
		int lib;
		for (uint64_t g = g0; g < g1 && cur_next(&cu, &x, &lib); g++) {
			const uint64_t w = x.b->info[x.i];
			const int nh = (int)((w >> 40) & 255);
			if ((w >> 57) & 1) over++;
			if (!nh) continue;
			mapped++;
			const unsigned long long rc = g + 1;                                     /* readCounter */
			const sdt_hit *more = &x.b->hits[w & ((1ULL << 40) - 1)];               /* hits 1.. of the read; hit 0 = hits[read] */
#define HIT(m) ((m) == 0 ? &x.b->hits[x.i] : &more[(m) - 1])
			const sdt_hit *h = (rc % 2 == 1) ? HIT(nh - 1) : HIT(0);                 /* :566-569 */
			tb_room(ro, 64);
			tb_u64(ro, rc); ro->p[ro->n++] = '\t';
			tb_u64(ro, h->contig); ro->p[ro->n++] = '\t';
			tb_i64(ro, (long long)h->contig_offset - (long long)h->read_offset + 1); ro->p[ro->n++] = '\t';
			ro->p[ro->n++] = (h->align_len_orien >> 31) ? '-' : '+'; ro->p[ro->n++] = '\n';
			for (int m = 0; m < nh; m++) {
				const sdt_hit *hm = HIT(m);
				const int al = (int)(hm->align_len_orien & 0x7FFFFFFFu);
				const char orien = (hm->align_len_orien >> 31) ? '-' : '+';
				if (al < 5) continue;
				tb_room(c2, 64);
				tb_u64(c2, rc); c2->p[c2->n++] = '\t';
				tb_u64(c2, hm->contig); c2->p[c2->n++] = '\t';
				tb_i64(c2, (long long)hm->read_offset - (long long)hm->contig_offset); c2->p[c2->n++] = '\t';
				c2->p[c2->n++] = orien; c2->p[c2->n++] = '\n';
				if (!F->read_trace) continue;
				const int span = al + F->K - 1;                                       /* :573-582 */
				tb_room(ri, 128);
				tb_u64(ri, rc); ri->p[ri->n++] = '\t';
				tb_i64(ri, (long long)hm->read_offset - 1); ri->p[ri->n++] = '\t';
				if (orien == '+') {
					tb_u64(ri, hm->contig); ri->p[ri->n++] = '\t';
					tb_i64(ri, hm->contig_offset); ri->p[ri->n++] = '\t';
				} else {
					tb_u64(ri, F->ctg_twin[hm->contig]); ri->p[ri->n++] = '\t';
					tb_i64(ri, (long long)(int)F->ctg_len[hm->contig] - hm->contig_offset - span); ri->p[ri->n++] = '\t';
				}
				tb_i64(ri, span); ri->p[ri->n++] = '\t';
				ri->p[ri->n++] = orien; ri->p[ri->n++] = '\n';
			}
		}
		F->ro[bi] = ro_l; F->c2[bi] = c2_l; F->ri[bi] = ri_l;
		F->mapped[bi] = mapped;
		F->overflowed[bi] = over;
	}
}

typedef struct { fmt_ctx *F; int fd[3]; off_t *off[3]; volatile int failed; } pw_ctx;

static void pwrite_all(pw_ctx *W, int fd, const char *p, size_t n, off_t off)
{
	while (n) {
		const ssize_t w = pwrite(fd, p, n, off);
		if (w <= 0) { W->failed = 1; return; }
		p += w; n -= (size_t)w; off += w;
	}
}

static void pwrite_blocks(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	pw_ctx *W = (pw_ctx *)vc;
	for (uint64_t b = lo; b < hi; b++) {
		pwrite_all(W, W->fd[0], W->F->ro[b].p, W->F->ro[b].n, W->off[0][b]);
		pwrite_all(W, W->fd[1], W->F->c2[b].p, W->F->c2[b].n, W->off[1][b]);
		if (W->fd[2] >= 0) pwrite_all(W, W->fd[2], W->F->ri[b].p, W->F->ri[b].n, W->off[2][b]);
	}
}

static void usage(void)
{
	printf("\nmap -s configFile -g inputGraph [-p n_cpu -K kmer -r]\n");
	printf("  -s <string>      configFile: the config file of solexa reads\n");
	printf("  -g <string>      inputGraph: prefix of input graph file names\n");
	printf("  -p <int>         n_cpu: number of cpu for use, [8]\n");
	printf("  -K <int>         kmer(min 13, max 127): kmer size, [23]\n");
	printf("  -f (optional)    output gap related reads for SRkgf to fill gap, [NO]\n");
	printf("  -r (optional)    output the information between read and scaffold, [NO]\n");
}

/* writeChar2tightString (seq.c:49-71) */
static inline void tight_put(unsigned nt, char *tight, int pos)
{
	char *byte = tight + pos / 4;
	switch (pos % 4) {
	case 0: *byte &= 63; *byte += (char)(nt << 6); return;
	case 1: *byte &= (char)207; *byte += (char)(nt << 4); return;
	case 2: *byte &= (char)243; *byte += (char)(nt << 2); return;
	default: *byte &= (char)252; *byte += (char)nt; return;
	}
}

typedef struct { FILE *gap; char *rc1; long long reads_in_gap; gzFile fill_gap, fill_pe; } gap_out;

/* output1read (prlRead2Ctg.c:423-446) */
static void output1read(gap_out *G, rref x, int ctg, int pos, char orien, int ins, int dhflag)
{
	const int len = rlen(x);
	G->reads_in_gap++;
	for (int i = 0; i < len; i++) tight_put(rbase(x, i), G->rc1, i);
	fwrite(&len, sizeof(int), 1, G->gap);
	fwrite(&ctg, sizeof(int), 1, G->gap);
	fwrite(&pos, sizeof(int), 1, G->gap);
	fwrite(G->rc1, 1, (size_t)(len / 4 + 1), G->gap);
	if (G->fill_gap && ins < 2000 && len > 0) {
		gzprintf(G->fill_gap, ">%d\t%d\t%d\t%c\t%d\t%d\n", len, ctg, pos, orien, ins, dhflag);
		for (int i = 0; i < len; i++) gzputc(G->fill_gap, "ACTG"[rbase(x, i)]);
		gzputc(G->fill_gap, '\n');
	}
}

/* getPEreadOnContig (:493-524): both mates on contigs; their tight strings pass through the same shared buffer */
static void pe_half(gap_out *G, rref x, int ctg, int pos, char orien, int ins)
{
	const int len = rlen(x);
	gzwrite(G->fill_pe, &len, sizeof(int));
	gzwrite(G->fill_pe, &ctg, sizeof(int));
	gzwrite(G->fill_pe, &pos, sizeof(int));
	gzwrite(G->fill_pe, &orien, 1);
	gzwrite(G->fill_pe, &ins, sizeof(int));
	for (int i = 0; i < len; i++) tight_put(rbase(x, i), G->rc1, i);
	gzwrite(G->fill_pe, G->rc1, (unsigned)(len / 4 + 1));
}

int main(int argc, char **argv)
{
	char cfgfile[4096] = "", graph[4096] = "";
	int threads = 8, max_k = 0, device = 0, read_trace = 0, fill = 0, have_s = 0, have_g = 0, c;
	static struct option longopts[] = {{"max-k", required_argument, 0, 1000}, {"device", required_argument, 0, 1001},
	                                   {"batch-kmers", required_argument, 0, 1002}, {0, 0, 0, 0}};
	int batch_kmers = 100000000;                           /* buffer_size of prlRead2Ctg.c:31; --batch-kmers exists for the tests */
	if (argc > 1 && strcmp(argv[1], "map") == 0) { argv++; argc--; }
	while ((c = getopt_long(argc, argv, "s:g:K:p:rfR", longopts, NULL)) != -1) {
		switch (c) {
		case 's': have_s = 1; snprintf(cfgfile, sizeof cfgfile, "%s", optarg); break;
		case 'g': have_g = 1; snprintf(graph, sizeof graph, "%s", optarg); break;
		case 'K': break;                                   /* overwritten by <g>.preGraphBasic (map.c:71) */
		case 'p': threads = atoi(optarg); break;
		case 'r': read_trace = 1; break;
		case 'R': break;                                   /* RPKM: used by later stages only */
		case 'f': fill = 1; break;
		case 1000: max_k = atoi(optarg); break;
		case 1001: device = atoi(optarg); break;
		case 1002: batch_kmers = atoi(optarg); break;
		default: usage(); return 255;
		}
	}
	if (!have_s || !have_g) { usage(); return 255; }
	if (threads < 1) threads = 1;
	time_t t_start = time(NULL);
	g_t_last = now_ms();

	/* getMinOverlap (map.c:33-60) */
	char name[4400];
	int K = 23;
	snprintf(name, sizeof name, "%s.preGraphBasic", graph);
	FILE *fp = fopen(name, "r");
	if (fp) {
		char line[1024], ch;
		int nk;
		while (fgets(line, sizeof line, fp))
			if (line[0] == 'V') sscanf(line + 6, "%d %c %d", &nk, &ch, &K);
		fclose(fp);
	}
	if (max_k == 0) max_k = K <= 31 ? 31 : (K <= 63 ? 63 : 127);
	printf("K = %d\n", K);
	const int ctg_short = K + 2;
	printf("contig len cutoff: %d\n", ctg_short);

	/* ---- prlContig2nodes ---- */
	time_t t0 = time(NULL);
	contigs_t C;
	snprintf(name, sizeof name, "%s.contig", graph);
	if (load_contigs(name, K, ctg_short, &C) != 0) return 255;
	printf("\nthere're %lld contigs in file: %s, max seq len %d, min seq len %d, max name len %d\n", C.num_seq, graph, C.max_len, C.min_len, C.name_len);
	printf("time spent on parse contigs file %ds\n", (int)(time(NULL) - t0));
	phase("parse contigs");
	sdt_ctx *gpu = NULL;
	uint64_t ctg_kmers = 0;
	for (uint64_t i = 0; i < C.n; i++) ctg_kmers += C.offs[i + 1] - C.offs[i] - (uint64_t)K + 1;
	if (sdt_gpu_init(&gpu, device, K, ctg_kmers + 1024, SDT_FLAG_CONTIG_INDEX) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_init: %s\n", sdt_gpu_last_error());
		return 1;
	}
	printf("%d thread created in prlHashCtg\n", threads);
	t0 = time(NULL);
	uint64_t kmers = 0, nodes = 0;
	if (sdt_gpu_index_contigs(gpu, C.words, C.nwords, C.offs, C.ids, C.n) != SDT_OK || sdt_gpu_finish_count(gpu, &kmers, &nodes) != SDT_OK) {
		fprintf(stderr, "contig index: %s\n", sdt_gpu_last_error());
		return 1;
	}
	printf("time spent on hash reads: %ds\n", (int)(time(NULL) - t0));
	printf("%lli nodes allocated, %lli kmer in reads, %lli kmer processed\n", (long long)nodes, (long long)kmers, (long long)kmers);
	printf("time spent on De bruijn graph construction: %ds\n\n", (int)(time(NULL) - t0));
	phase("contig index (GPU)");

	/* ---- prlRead2Ctg ---- */
	t0 = time(NULL);
	sdt_cfg cfg;
	if (sdt_cfg_load(cfgfile, &cfg) != 0) return 255;
	const int max_read_len = cfg.max_rd_len ? cfg.max_rd_len : 100;
	printf("In file: %s, max seq len %d, max name len %d\n\n", cfgfile, max_read_len, 256);
	printf("%d thread created in prlRead2Ctg\n", threads);
	/* basicContigInfo (prlRead2Ctg.c:610-648) */
	snprintf(name, sizeof name, "%s.ContigIndex", graph);
	fp = fopen(name, "r");
	if (!fp) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	char line[1024];
	int num_all = 0, num_long = 0;
	if (!fgets(line, sizeof line, fp)) { fclose(fp); return 255; }
	sscanf(line + 8, "%d %d", &num_all, &num_long);
	printf("%d edges in graph\n", num_all);
	uint32_t *ctg_len = (uint32_t *)calloc((size_t)num_all + 2, sizeof(uint32_t)), *ctg_twin = (uint32_t *)calloc((size_t)num_all + 2, sizeof(uint32_t));
	if (!fgets(line, sizeof line, fp)) { fclose(fp); return 255; }
	for (int k = 0; fgets(line, sizeof line, fp);) {
		int index, length, bal;
		if (sscanf(line, "%d %d %d", &index, &length, &bal) != 3) continue;
		if (k + 1 > num_all) break;
		k++;
		ctg_len[k] = (uint32_t)length;
		ctg_twin[k] = (uint32_t)(k + (bal + 1) - 1);                              /* getTwinCtg, attachPEinfo.c:479-482 */
		if (index != k) printf("basicContigInfo: %d vs %d\n", index, k);
		if (bal == 0 || k + 1 > num_all) continue;
		k++;
		ctg_len[k] = (uint32_t)length;
		ctg_twin[k] = (uint32_t)(k + (-bal + 1) - 1);
	}
	fclose(fp);
	if (sdt_gpu_set_contig_table(gpu, ctg_len, ctg_twin, (uint64_t)num_all) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }

	/* every paired input of the asm_flags 2|3 libraries, in read1seqInLib's order (readseq1by1.c:557-636,935-1131) */
	mstream *S = NULL;
	int ns = 0, scap = 0;
	uint64_t total = 0;
	const size_t chunk = 32u << 20;
	for (int i = 0; i < cfg.nlibs; i++) {
		const sdt_lib *l = &cfg.libs[i];
		if (l->asm_flag != 2 && l->asm_flag != 3) continue;
		int mrl = max_read_len;
		if (l->rd_len_cutoff > 0 && l->rd_len_cutoff < mrl) mrl = l->rd_len_cutoff;
		if (l->nb) { fprintf(stderr, "b= (BAM) input is not supported by this build\n"); return 1; }
		struct { char **a; char **b; int n; int fmt; int type; } groups[] = {
			{l->f1, l->f2, l->nf1 < l->nf2 ? l->nf1 : l->nf2, 'a', 1}, {l->q1, l->q2, l->nq1 < l->nq2 ? l->nq1 : l->nq2, 'q', 2}, {l->p, NULL, l->np, 'a', 3}};
		for (unsigned g = 0; g < 3; g++)
			for (int f = 0; f < groups[g].n; f++) {
				if (ns == scap) { scap = scap ? scap * 2 : 16; S = (mstream *)realloc(S, (size_t)scap * sizeof(mstream)); }
				mstream *s = &S[ns++];
				memset(s, 0, sizeof *s);
				s->lib = i;
				s->paired = groups[g].b != NULL;
				s->base = total;
				if (sdt_read_file(groups[g].a[f], groups[g].fmt, mrl, l->reverse, threads, chunk, keep_batch, &s->A, NULL) != 0) return 1;
				if (s->paired && sdt_read_file(groups[g].b[f], groups[g].fmt, mrl, l->reverse, threads, chunk, keep_batch, &s->B, NULL) != 0) return 1;
				if (s->paired && s->A.nreads != s->B.nreads) {
					fprintf(stderr, "%s and %s hold %llu and %llu reads: paired files must have the same number of records\n", groups[g].a[f],
					        groups[g].b[f], (unsigned long long)s->A.nreads, (unsigned long long)s->B.nreads);
					return 1;
				}
				index_file(&s->A);
				if (s->paired) index_file(&s->B);
				s->nreads = s->A.nreads + s->B.nreads;
				total += s->nreads;
			}
	}
	phase("parse reads");

	/* ALIGNLEN: a global the main thread keeps updating while it fills a batch; parse1read sees what the LAST read
	 * of the batch left (prlRead2Ctg.c:774-805).  maxReadNum reads per batch (:689-690). */
	int max_read_num = batch_kmers / (max_read_len - K + 1 > 0 ? max_read_len - K + 1 : 1);
	if (max_read_num % 2) max_read_num--;
	if (max_read_num < 2) max_read_num = 2;
	const uint64_t nbatch = total ? (total + (uint64_t)max_read_num - 1) / (uint64_t)max_read_num : 0;
	int *batch_alen = (int *)calloc(nbatch + 1, sizeof(int));
	{
		cursor cu = {S, ns, 0, 0, 0, 0, 0, 0};
		rref x;
		int lib, prev_lib = -1, align_len = 0;
		for (uint64_t g = 0; cur_next(&cu, &x, &lib); g++) {
			const int ins = cfg.libs[lib].avg_ins;
			if (lib != prev_lib) {
				prev_lib = lib;
				align_len = cfg.libs[lib].map_len;
				if (ins > 1000) align_len = align_len < 35 ? 35 : align_len;
				else align_len = align_len < 32 ? 32 : align_len;
			}
			if (ins > 1000) { const int h = rlen(x) / 2 + 1; if (align_len < h) align_len = h; }
			batch_alen[g / (uint64_t)max_read_num] = align_len;
		}
		cursor c2 = {S, ns, 0, 0, 0, 0, 0, 0};
		for (uint64_t g = 0; cur_next(&c2, &x, &lib); g++) {
			if (!x.b->alen) x.b->alen = (int32_t *)malloc((x.b->nreads + 1) * sizeof(int32_t));
			x.b->alen[x.i] = batch_alen[g / (uint64_t)max_read_num];
		}
	}
	phase("ALIGNLEN per batch");

	/* chopKmer4read + searchKmer + parse1read on the GPU, chunk by chunk */
	for (int si = 0; si < ns; si++)
		for (int fb = 0; fb < 2; fb++) {
			mfile *f = fb ? &S[si].B : &S[si].A;
			for (size_t bi = 0; bi < f->n; bi++) {
				mbatch *m = &f->b[bi];
				if (!m->nreads) continue;
				m->info = (uint64_t *)malloc(m->nreads * sizeof(uint64_t));
				uint64_t cap = m->nreads + m->nreads / 8 + 64, got = 0;       /* first hit of every read + the further hits */
				for (;;) {
					m->hits = (sdt_hit *)malloc(cap * sizeof(sdt_hit));
					const int rc = sdt_gpu_align_reads(gpu, m->words, m->nwords, m->offs, m->nreads, m->alen, 0, m->info, m->hits, cap, &got);
					if (rc == SDT_OK) break;
					if (rc == SDT_EFULL && got > cap) { free(m->hits); cap = got; continue; }
					fprintf(stderr, "sdt_gpu_align_reads: %s\n", sdt_gpu_last_error());
					return 1;
				}
				m->nhits = got;
			}
		}
	phase("align reads (GPU)");

	/* ---- recordAlldgn (prlRead2Ctg.c:526-608) ---- */
	snprintf(name, sizeof name, "%s.readInGap", graph);
	gap_out G = {fopen(name, "wb"), (char *)calloc((size_t)max_read_len + 8, 1), 0, NULL, NULL};
	if (!G.gap) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	snprintf(name, sizeof name, "%s.readOnContig", graph);
	FILE *fo = fopen(name, "w");
	if (!fo) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	snprintf(name, sizeof name, "%s.ctg2Read", graph);
	FILE *f3 = fopen(name, "w");
	if (!f3) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	if (fill) {
		snprintf(name, sizeof name, "%s.shortreadInGap.gz", graph);
		G.fill_gap = gzopen(name, "w");
		snprintf(name, sizeof name, "%s.PEreadOnContig.gz", graph);
		G.fill_pe = gzopen(name, "wb");
		if (!G.fill_gap || !G.fill_pe) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	}
	FILE *f4 = NULL;
	if (read_trace) {
		snprintf(name, sizeof name, "%s.readInformation", graph);
		f4 = fopen(name, "w");
		if (!f4) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	}
	fprintf(fo, "read\tcontig\tpos\n");
	fprintf(f3, "read\tcontig\tpos\n");
	long long map_counter = 0, overflowed = 0;
	{
		/* the text files depend on each read alone: format blocks of reads in parallel, write them in order */
		const int nt = par_threads();
		const uint64_t block = 1 << 18;
		const uint64_t nblocks = (total + block - 1) / block;
		double t_fmt = 0, t_wr = 0;
		fflush(fo); fflush(f3);
		off_t pos_ro = ftello(fo), pos_c2 = ftello(f3), pos_ri = 0;
		fmt_ctx F = {S, ns, K, read_trace, ctg_len, ctg_twin, total, block, 0, (tbuf *)calloc((size_t)nt * 2, sizeof(tbuf)),
		             (tbuf *)calloc((size_t)nt * 2, sizeof(tbuf)), (tbuf *)calloc((size_t)nt * 2, sizeof(tbuf)),
		             (long long *)calloc((size_t)nt * 2, sizeof(long long)), (long long *)calloc((size_t)nt * 2, sizeof(long long))};
		for (uint64_t first = 0; first < nblocks; first += (uint64_t)nt * 2) {
			const uint64_t n = nblocks - first < (uint64_t)nt * 2 ? nblocks - first : (uint64_t)nt * 2;
			F.wave_first = first;
			const double tf0 = now_ms();
			par_for(0, n, 1, fmt_blocks, &F);
			t_fmt += now_ms() - tf0;
			/* every block knows its size now: give it its place in the file and let the threads write side by side */
			fflush(fo); fflush(f3);
			if (f4) fflush(f4);
			pw_ctx W = {&F, {fileno(fo), fileno(f3), f4 ? fileno(f4) : -1}, {(off_t *)malloc(n * sizeof(off_t)), (off_t *)malloc(n * sizeof(off_t)), (off_t *)malloc(n * sizeof(off_t))}, 0};
			for (uint64_t b = 0; b < n; b++) {
				W.off[0][b] = pos_ro; pos_ro += (off_t)F.ro[b].n;
				W.off[1][b] = pos_c2; pos_c2 += (off_t)F.c2[b].n;
				W.off[2][b] = pos_ri; pos_ri += (off_t)F.ri[b].n;
				map_counter += F.mapped[b];
				overflowed += F.overflowed[b];
			}
			const double tw0 = now_ms();
			par_for(0, n, 1, pwrite_blocks, &W);
			t_wr += now_ms() - tw0;
			free(W.off[0]); free(W.off[1]); free(W.off[2]);
			if (W.failed) { printf("write error on the read-to-contig files\n"); return 255; }
		}
		for (int b = 0; b < nt * 2; b++) { free(F.ro[b].p); free(F.c2[b].p); free(F.ri[b].p); }
		free(F.ro); free(F.c2); free(F.ri); free(F.mapped); free(F.overflowed);
		if (sdt_env("SDT_TIMING")) fprintf(stderr, "[sdt-map]   format %.1f ms, pwrite %.1f ms\n", t_fmt, t_wr);
	}
	phase("text files (parallel format)");
	long long read_counter = 0;
	{
		cursor cu = {S, ns, 0, 0, 0, 0, 0, 0};
		rref x, prev = {NULL, 0};
		int lib, prev_lib_seen = -1, prev_stream = -1;
		int ctg_prev = 0, pos_prev = 0, foot_prev = 0, lib_prev = 0;
		/* orienArray: written for mapped reads only, so an unmapped read shows what an earlier batch left at its index */
		char *orien_at = (char *)calloc((size_t)max_read_num + 2, 1);
		for (uint64_t g = 0; cur_next(&cu, &x, &lib); g++) {
			/* the lines the reference prints while it opens files and switches libraries */
			if (cu.si != prev_stream) {
				prev_stream = cu.si;
				const sdt_lib *l = &cfg.libs[lib];
				const mstream *s = &S[cu.si];
				/* which file pair of the library this is: count the earlier streams of the same library */
				int nth = 0;
				for (int q = 0; q < cu.si; q++) if (S[q].lib == lib) nth++;
				const int n1 = l->nf1 < l->nf2 ? l->nf1 : l->nf2, n2 = l->nq1 < l->nq2 ? l->nq1 : l->nq2;
				if (nth < n1) printf("read from file - type 1:\n %s\nread from file - type 1:\n %s\n", l->f1[nth], l->f2[nth]);
				else if (nth < n1 + n2) printf("read from file - type 2:\n %s\nread from file - type 2:\n %s\n", l->q1[nth - n1], l->q2[nth - n1]);
				else printf("read from file - type 3:\n %s\n", l->p[nth - n1 - n2]);
				(void)s;
			}
			if (lib != prev_lib_seen) {
				prev_lib_seen = lib;
				printf("current insert size %d, map_len %d\n", cfg.libs[lib].avg_ins, cfg.libs[lib].map_len);
			}
			const uint64_t t = g % (uint64_t)max_read_num;                             /* index inside the batch */
			if (t == 0) {
				/* signal 2: thread 0 chops reads 0, p, 2p, ... of the batch and leaves each one's reverse complement in
				 * rcSeq[1], the buffer output1read then packs reads into.  Later reads overwrite earlier ones, so walk
				 * back from the last and fill only what is not covered yet. */
				uint64_t rc = total - g < (uint64_t)max_read_num ? total - g : (uint64_t)max_read_num;
				int covered = 0;
				for (uint64_t tt = (rc - 1) / (uint64_t)threads * (uint64_t)threads;; tt -= (uint64_t)threads) {
					const rref y = global_read(S, ns, g + tt);
					const int len = rlen(y);
					if (len >= K + 1 && len > covered) {
						for (int i = covered; i < len; i++) G.rc1[i] = (char)(rbase(y, len - 1 - i) ^ 2u);
						covered = len;
					}
					if (covered >= max_read_len || tt < (uint64_t)threads) break;
				}
			}
			read_counter++;
			const uint64_t w = x.b->info[x.i];
			const int nh = (int)((w >> 40) & 255), best = (int)((w >> 48) & 255), foot = (int)((w >> 56) & 1);
			const sdt_hit *hb = !nh ? NULL : (best == 0 ? &x.b->hits[x.i] : &x.b->hits[(w & ((1ULL << 40) - 1)) + (uint64_t)best - 1]);
			int ctg = nh ? (int)hb->contig : 0;
			int pos = nh ? hb->contig_offset - (int)hb->read_offset + 1 : 0;
			if (nh) orien_at[t] = (hb->align_len_orien >> 31) ? '-' : '+';
			const int ins = cfg.libs[lib].avg_ins, ins_prev = cfg.libs[lib_prev].avg_ins;
			const int ctg_at_top = ctg;
			int rd1gap = 0, rd2gap = 0;
			if (t % 2 == 1 && prev.b) {
				if (ctg < 1 && ctg_prev > 0) {                                         /* read 2 in gap (:541-545, getReadIngap) */
					ctg = ctg_prev;
					pos = pos_prev + ins - rlen(x);
					output1read(&G, x, ctg, pos, orien_at[t - 1] == '+' ? '-' : '+', ins, 1);
					rd2gap = 1;
				} else if (ctg > 0 && ctg_prev < 1) {                                  /* read 1 in gap */
					ctg_prev = ctg;
					pos_prev = pos + ins_prev - rlen(prev);
					output1read(&G, prev, ctg_prev, pos_prev, orien_at[t] == '+' ? '-' : '+', ins_prev, 1);
					rd1gap = 1;
				} else if (ctg > 0 && ctg_prev > 0 && fill && ins < 2000 && ins == ins_prev) {   /* :554-558, 504 */
					pe_half(&G, prev, ctg_prev, pos_prev, orien_at[t - 1], ins_prev);
					pe_half(&G, x, ctg, pos, orien_at[t], ins);
				}
			}
			if (ctg_at_top >= 1) {
				if (t % 2 == 1 && prev.b) {
					/* "reads are not located by pe info but across edges" (:591-606); a footprint read is always mapped, so
					 * locate1read is never reached */
					if (foot_prev && !rd1gap) output1read(&G, prev, ctg_prev, pos_prev, orien_at[t] == '+' ? '-' : '+', ins_prev, 1);
					if (foot && !rd2gap) output1read(&G, x, ctg, pos, orien_at[t - 1] == '+' ? '-' : '+', ins, 2);
				}
			}
			prev = x;
			ctg_prev = ctg; pos_prev = pos; foot_prev = foot; lib_prev = lib;
			if (t % 2 == 1) prev.b = NULL;                                             /* pairs never straddle (t-1, t) with t even */
		}
		free(orien_at);
	}
	if (total % (uint64_t)max_read_num)                                               /* printed only when the last batch was not empty (:813-821) */
		printf("Output %lld out of %lld (%.1f)%% reads in gaps\n", G.reads_in_gap, read_counter, (float)G.reads_in_gap / read_counter * 100);
	printf("%lld out of %lld (%.1f)%% reads mapped to contigs\n", map_counter, read_counter, (float)map_counter / read_counter * 100);
	if (overflowed)
		fprintf(stderr, "%lld reads touch more than 20 contigs with >= ALIGNLEN-K+1 k-mers each (the reference overruns a 20-entry array there); reported unmapped\n", overflowed);
	fclose(fo); fclose(f3); fclose(G.gap);
	if (f4) fclose(f4);
	if (G.fill_gap) gzclose(G.fill_gap);
	if (G.fill_pe) gzclose(G.fill_pe);
	phase("readInGap (ordered pass)");
	/* *.peGrads (:825-846): one line per library that delivered reads, boundaries in reads */
	snprintf(name, sizeof name, "%s.peGrads", graph);
	fo = fopen(name, "w");
	if (!fo) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
	int grads = 0;
	for (int i = 0; i < cfg.nlibs; i++) {
		uint64_t n = 0;
		for (int q = 0; q < ns; q++) if (S[q].lib == i) n += S[q].nreads;
		if (n) grads++;
	}
	fprintf(fo, "grads&num: %d\t%lld\t%d\n", grads, (long long)total, max_read_len);
	uint64_t bound = 0;
	for (int i = 0; i < cfg.nlibs; i++) {
		uint64_t n = 0;
		for (int q = 0; q < ns; q++) if (S[q].lib == i) n += S[q].nreads;
		if (!n) continue;
		bound += n;
		fprintf(fo, "%d\t%lld\t%d\t%d\n", cfg.libs[i].avg_ins, (long long)bound, cfg.libs[i].rank, cfg.libs[i].pair_num_cut);
	}
	fclose(fo);
	if (grads) printf("%d pe insert size, the largest boundary is %lld\n\n", grads, (long long)bound);
	else printf("no paired reads found\n");
	for (int i = 0; i < cfg.nlibs; i++) printf("[LIB] %d, avg_ins %d, reverse %d \n", i, cfg.libs[i].avg_ins, cfg.libs[i].reverse);   /* free_libs */
	printf("time spent on mapping reads: %ds\n\n", (int)(time(NULL) - t0));
	printf("overall time for alignment: %dm\n\n", (int)(time(NULL) - t_start) / 60);
	sdt_gpu_destroy(gpu);
	sdt_cfg_free(&cfg);
	return 0;
}
