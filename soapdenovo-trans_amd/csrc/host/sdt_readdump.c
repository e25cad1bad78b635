/* sdt_readdump.c -- host-logic self test (no GPU): run the config parser + the parallel reader/packer exactly
 * as sdt-pregraph does and print every read back as text (one line per read, ACTG alphabet of inc/def.h:40),
 * so tests can compare the ingest with the oracle's restatement of readseqfq / readseqInBuf. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libcfg.h"
#include "seqio.h"

/* SDT_READDUMP_POOL=<n>: the reader packs into a pool of n buffers (seqio.h) and this consumer keeps every batch's buffer
 * for three more batches before it gives it back -- the way sdt-pregraph's asynchronous pushes hold theirs */
static int g_held[3] = {-1, -1, -1}, g_nheld = 0;

static int dump(void *user, const sdt_batch *b)
{
	FILE *fo = (FILE *)user;
	if (b->pool_slot >= 0) {
		if (g_nheld == 3) {
			sdt_pool_release(g_held[0]);
			g_held[0] = g_held[1]; g_held[1] = g_held[2];
			g_nheld = 2;
		}
		sdt_pool_take(b->pool_slot);
		g_held[g_nheld++] = b->pool_slot;
	}
	if (b->fixed_len)                                     /* the fixed-length form must describe the same reads */
		for (uint64_t r = 0; r <= b->nreads; r++)
			if (b->offsets[r] != r * b->fixed_len) { fprintf(stderr, "fixed_len %llu but offsets[%llu] = %llu\n", (unsigned long long)b->fixed_len, (unsigned long long)r, (unsigned long long)b->offsets[r]); return -1; }
	for (uint64_t r = 0; r < b->nreads; r++) {
		for (uint64_t i = b->offsets[r]; i < b->offsets[r + 1]; i++)
			fputc("ACTG"[(b->words[i >> 4] >> (30 - 2 * (i & 15))) & 3], fo);
		fputc('\n', fo);
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: sdt-readdump <cfg> [threads] [chunk_bytes]\n"); return 2; }
	int threads = argc > 2 ? atoi(argv[2]) : 4;
	size_t chunk = argc > 3 ? (size_t)atol(argv[3]) : (32u << 20);
	sdt_cfg cfg;
	if (sdt_cfg_load(argv[1], &cfg) != 0) return 1;
	if (getenv("SDT_READDUMP_POOL")) sdt_pool_enable(malloc, free, atoi(getenv("SDT_READDUMP_POOL")));
	int max_read_len = cfg.max_rd_len ? cfg.max_rd_len : 100;
	printf("#libs %d max_rd_len %d\n", cfg.nlibs, max_read_len);
	for (int i = 0; i < cfg.nlibs; i++) {
		sdt_lib *l = &cfg.libs[i];
		printf("#lib avg_ins %d asm_flag %d reverse %d rd_len_cutoff %d\n", l->avg_ins, l->asm_flag, l->reverse, l->rd_len_cutoff);
		if (l->asm_flag != 1 && l->asm_flag != 3) continue;
		int mrl = max_read_len;
		if (l->rd_len_cutoff > 0 && l->rd_len_cutoff < mrl) mrl = l->rd_len_cutoff;
		struct { char **names; int n; int fmt; } groups[] = {{l->f1, l->nf1, 'a'}, {l->f2, l->nf2, 'a'}, {l->q1, l->nq1, 'q'},
			{l->q2, l->nq2, 'q'}, {l->p, l->np, 'a'}, {l->f, l->nf, 'a'}, {l->q, l->nq, 'q'}};
		for (unsigned g = 0; g < sizeof groups / sizeof groups[0]; g++)
			for (int f = 0; f < groups[g].n; f++) {
				printf("#file %s\n", groups[g].names[f]);
				if (sdt_read_file(groups[g].names[f], groups[g].fmt, mrl, l->reverse, threads, chunk, dump, stdout, NULL) != 0)
					return 1;
				while (g_nheld) sdt_pool_release(g_held[--g_nheld]);
			}
	}
	sdt_cfg_free(&cfg);
	return 0;
}
