/* sdt_readdump.c -- host-logic self test (no GPU): run the config parser + the parallel reader/packer exactly
 * as sdt-pregraph does and print every read back as text (one line per read, ACTG alphabet of inc/def.h:40),
 * so tests can compare the ingest with the oracle's restatement of readseqfq / readseqInBuf. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../sdt_knobs.h"
#include "libcfg.h"
#include "seqio.h"
#include "readstream.h"

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

/* --ordinals <nranks>: the multi-process reader without a GPU.  The pass is walked once as one rank (the ordinals sdt_stream_reads
 * hands out are the truth), then once per rank of an N-rank job with foreign chunks skipped (seqio.h: sdt_read_shard_skip_foreign):
 * every chunk must be parsed by exactly its owner, nobody may read a byte of a foreign chunk, and the ordinals worked out from the
 * owners' record counts (readstream.h: sdt_stream_ordinals, what sdt-pregraph does after its all-gather per group) must be the truth. */
typedef struct { uint64_t ord, n; int sid, stride, parity, owner, unknown; } chunk_rec;
typedef struct { chunk_rec *v; size_t n, cap; } chunk_log;

static int log_chunk(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride)
{
	chunk_log *L = (chunk_log *)user;
	if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 256; L->v = (chunk_rec *)realloc(L->v, L->cap * sizeof(chunk_rec)); }
	if (b->chunk_index != L->n) { fprintf(stderr, "chunk %zu reported as %llu\n", L->n, (unsigned long long)b->chunk_index); return -1; }
	chunk_rec r = {ord_base, b->nreads, b->stream_id, (int)ord_stride, b->stream_parity, b->owner, b->count_unknown};
	L->v[L->n++] = r;
	return 0;
}

static int check_ordinals(const sdt_cfg *cfg, int max_read_len, int threads, size_t chunk, int nranks)
{
	chunk_log truth = {NULL, 0, 0};
	sdt_read_shard_begin(0, 1, 0);
	if (sdt_stream_reads(cfg, max_read_len, threads, chunk, 0, log_chunk, &truth, NULL) != 0) return 1;
	chunk_log *per = (chunk_log *)calloc((size_t)nranks, sizeof(chunk_log));
	uint64_t parsed = 0, seen = 0;
	for (int r = 0; r < nranks; r++) {
		sdt_read_shard_begin(r, nranks, 0);
		sdt_read_shard_skip_foreign(1);
		if (sdt_stream_reads(cfg, max_read_len, threads, chunk, 0, log_chunk, &per[r], NULL) != 0) return 1;
		if (per[r].n != truth.n) { fprintf(stderr, "rank %d saw %zu chunks, one rank sees %zu\n", r, per[r].n, truth.n); return 1; }
		parsed += sdt_reader_bytes_parsed;
		seen = sdt_reader_bytes_seen;
	}
	sdt_read_shard_begin(0, 1, 0);
	if (parsed != seen) { fprintf(stderr, "the ranks parsed %llu bytes in all, the input has %llu\n", (unsigned long long)parsed, (unsigned long long)seen); return 1; }
	sdt_stream_ordinals so;
	sdt_stream_ordinals_init(&so);
	for (size_t c = 0; c < truth.n; c++) {
		const int owner = (int)(c % (size_t)nranks);
		for (int r = 0; r < nranks; r++) {
			const chunk_rec *x = &per[r].v[c];
			if (x->owner != owner || x->unknown != (r != owner) || (r != owner && x->n != 0)) { fprintf(stderr, "chunk %zu on rank %d: owner %d unknown %d n %llu\n", c, r, x->owner, x->unknown, (unsigned long long)x->n); return 1; }
			if (x->sid != truth.v[c].sid || x->stride != truth.v[c].stride || x->parity != truth.v[c].parity) { fprintf(stderr, "chunk %zu on rank %d: another stream\n", c, r); return 1; }
		}
		const chunk_rec *o = &per[owner].v[c];
		if (o->n != truth.v[c].n) { fprintf(stderr, "chunk %zu: its owner counts %llu records, one rank %llu\n", c, (unsigned long long)o->n, (unsigned long long)truth.v[c].n); return 1; }
		const uint64_t base = sdt_stream_ordinals_next(&so, o->sid, o->stride, o->parity, o->n);
		if (base != truth.v[c].ord) { fprintf(stderr, "chunk %zu: ordinal %llu worked out, %llu is the truth\n", c, (unsigned long long)base, (unsigned long long)truth.v[c].ord); return 1; }
	}
	printf("ordinals OK: %zu chunks, %d ranks, %llu bytes each parsed once\n", truth.n, nranks, (unsigned long long)seen);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc >= 4 && strcmp(argv[1], "--ordinals") == 0) {
		sdt_cfg cfg;
		if (sdt_cfg_load(argv[3], &cfg) != 0) return 1;
		const int rc = check_ordinals(&cfg, cfg.max_rd_len ? cfg.max_rd_len : 100, argc > 4 ? atoi(argv[4]) : 4, argc > 5 ? (size_t)atol(argv[5]) : 30000, atoi(argv[2]));
		sdt_cfg_free(&cfg);
		return rc;
	}
	if (argc < 2) { fprintf(stderr, "usage: sdt-readdump <cfg> [threads] [chunk_bytes]  |  sdt-readdump --ordinals <nranks> <cfg> [threads] [chunk_bytes]\n"); return 2; }
	int threads = argc > 2 ? atoi(argv[2]) : 4;
	size_t chunk = argc > 3 ? (size_t)atol(argv[3]) : (32u << 20);
	sdt_cfg cfg;
	if (sdt_cfg_load(argv[1], &cfg) != 0) return 1;
	if (sdt_test_env("SDT_READDUMP_POOL")) sdt_pool_enable(malloc, free, atoi(sdt_test_env("SDT_READDUMP_POOL")));
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
