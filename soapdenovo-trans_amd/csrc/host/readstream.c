/* readstream.c -- see readstream.h */
#include "readstream.h"
#include <stdio.h>

typedef struct {
	sdt_stream_fn fn;
	void *user;
	uint64_t ord, stride;
} relay_t;

static int relay(void *user, const sdt_batch *b)
{
	relay_t *r = (relay_t *)user;
	const int rc = r->fn(r->user, b, r->ord, r->stride);
	r->ord += b->nreads * r->stride;
	return rc;
}

int sdt_stream_reads(const sdt_cfg *cfg, int max_read_len, int threads, size_t chunk_bytes, int verbose,
                     sdt_stream_fn fn, void *user, uint64_t *nreads)
{
	uint64_t ordinal = 0;
	int rc = 0;
	for (int i = 0; i < cfg->nlibs && rc == 0; i++) {
		const sdt_lib *l = &cfg->libs[i];
		if (l->asm_flag != 1 && l->asm_flag != 3)
			continue;
		int mrl = max_read_len;                                              /* prlHashReads.c:820-823 */
		if (l->rd_len_cutoff > 0 && l->rd_len_cutoff < mrl) mrl = l->rd_len_cutoff;
		if (l->nb) {
			fprintf(stderr, "b= (BAM) input is not supported by this build\n");
			return -1;
		}
		struct { char **a; char **b; int n; int fmt; int type; } groups[] = {
			{l->f1, l->f2, l->nf1 < l->nf2 ? l->nf1 : l->nf2, 'a', 1}, {l->q1, l->q2, l->nq1 < l->nq2 ? l->nq1 : l->nq2, 'q', 2},
			{l->p, NULL, l->np, 'a', 3}, {l->f, NULL, l->nf, 'a', 5}, {l->q, NULL, l->nq, 'q', 6}};
		for (unsigned g = 0; g < sizeof groups / sizeof groups[0] && rc == 0; g++)
			for (int f = 0; f < groups[g].n && rc == 0; f++) {
				uint64_t n1 = 0, n2 = 0;
				if (verbose) {
					printf("read from file - type %d:\n %s\n", groups[g].type, groups[g].a[f]);
					if (groups[g].b) printf("read from file - type %d:\n %s\n", groups[g].type, groups[g].b[f]);
				}
				relay_t r = {fn, user, ordinal, groups[g].b ? 2u : 1u};
				rc = sdt_read_file(groups[g].a[f], groups[g].fmt, mrl, l->reverse, threads, chunk_bytes, relay, &r, &n1);
				if (rc == 0 && groups[g].b) {
					relay_t r2 = {fn, user, ordinal + 1, 2};
					rc = sdt_read_file(groups[g].b[f], groups[g].fmt, mrl, l->reverse, threads, chunk_bytes, relay, &r2, &n2);
				}
				ordinal += groups[g].b ? 2 * (n1 > n2 ? n1 : n2) : n1;
			}
	}
	if (nreads) *nreads = ordinal;
	return rc;
}
