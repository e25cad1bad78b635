/* readstream.c -- see readstream.h */
#include "readstream.h"
#include <stdio.h>

typedef struct {
	sdt_stream_fn fn;
	void *user;
	uint64_t ord, stride;
	int sid, parity;
} relay_t;

static int relay(void *user, const sdt_batch *b)
{
	relay_t *r = (relay_t *)user;
	sdt_batch bb = *b;
	bb.stream_id = r->sid;
	bb.stream_parity = r->parity;
	const int rc = r->fn(r->user, &bb, r->ord, r->stride);
	r->ord += b->nreads * r->stride;
	return rc;
}

/* the arithmetic of sdt_stream_reads below, one chunk at a time, for a caller that learns the record counts late (readstream.h) */
void sdt_stream_ordinals_init(sdt_stream_ordinals *s) { s->ordinal = 0; s->n_cur = 0; s->n_first = 0; s->sid = -1; s->stride = 1; s->parity = 0; s->open_pair = 0; }

uint64_t sdt_stream_ordinals_next(sdt_stream_ordinals *s, int stream_id, int stride, int parity, uint64_t nreads)
{
	if (stream_id != s->sid) {
		if (s->sid >= 0) {                                   /* the stream before this one is complete */
			if (s->stride == 1) s->ordinal += s->n_cur;
			else if (s->parity == 0) { s->n_first = s->n_cur; s->open_pair = 1; }      /* first file of a pair: the second one shares its base */
			else { s->ordinal += 2 * (s->n_first > s->n_cur ? s->n_first : s->n_cur); s->n_first = 0; s->open_pair = 0; }
		}
		/* a file without a single record yields no chunk and is never seen here: a pair whose second file was empty is closed now,
		 * a second file whose first was empty starts from nothing */
		const int second_of_open = s->open_pair && stride == 2 && parity == 1 && stream_id == s->sid + 1;
		if (s->open_pair && !second_of_open) { s->ordinal += 2 * s->n_first; s->n_first = 0; s->open_pair = 0; }
		if (parity == 1 && !second_of_open) s->n_first = 0;
		s->sid = stream_id; s->stride = stride; s->parity = parity; s->n_cur = 0;
	}
	const uint64_t base = s->ordinal + (uint64_t)parity + s->n_cur * (uint64_t)stride;
	s->n_cur += nreads;
	return base;
}

int sdt_stream_reads(const sdt_cfg *cfg, int max_read_len, int threads, size_t chunk_bytes, int verbose,
                     sdt_stream_fn fn, void *user, uint64_t *nreads)
{
	uint64_t ordinal = 0;
	int rc = 0, sid = 0;
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
				relay_t r = {fn, user, ordinal, groups[g].b ? 2u : 1u, sid++, 0};
				rc = sdt_read_file(groups[g].a[f], groups[g].fmt, mrl, l->reverse, threads, chunk_bytes, relay, &r, &n1);
				if (rc == 0 && groups[g].b) {
					relay_t r2 = {fn, user, ordinal + 1, 2, sid++, 1};
					rc = sdt_read_file(groups[g].b[f], groups[g].fmt, mrl, l->reverse, threads, chunk_bytes, relay, &r2, &n2);
				}
				ordinal += groups[g].b ? 2 * (n1 > n2 ? n1 : n2) : n1;
			}
	}
	if (nreads) *nreads = ordinal;
	return rc;
}
