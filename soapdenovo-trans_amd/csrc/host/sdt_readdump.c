/* sdt_readdump.c -- host-logic self test (no GPU): run the config parser + the parallel reader/packer exactly
 * as sdt-pregraph does and print every read back as text (one line per read, ACTG alphabet of inc/def.h:40),
 * so tests can compare the ingest with the oracle's restatement of readseqfq / readseqInBuf. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libcfg.h"
#include "seqio.h"

static int dump(void *user, const sdt_batch *b)
{
	FILE *fo = (FILE *)user;
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
			}
	}
	sdt_cfg_free(&cfg);
	return 0;
}
