/* graphcheck.c -- host-logic self test (no GPU): run the host graph phases of sdt-pregraph on a node dump.
 * dump (little endian): int32 K, nw_variant, nw_keys, p, d, dd; uint64 n; then keys[n*nw_keys] u64,
 * l_links[n] u32, r_flags[n] u32, count[n] u32, first[n] u64 -- exactly what sdt_gpu_export_nodes returns.
 * usage: sdt-graphcheck <dump> <out prefix> [<lib.cfg>]   (with a config: also the second read pass -> .preArc) */
#include <stdio.h>
#include <stdlib.h>
#include "../../sdt_knobs.h"
#include "graph.h"
#include <time.h>
#include "../readstream.h"

typedef struct { graph_t *G; struct arcs *A; } arc_state;
static int arc_batch(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride)
{
	arc_state *st = (arc_state *)user;
	uint8_t *codes = (uint8_t *)malloc(1 << 16);
	for (uint64_t r = 0; r < b->nreads; r++) {
		const uint64_t o = b->offsets[r], len = b->offsets[r + 1] - o;
		for (uint64_t i = 0; i < len && i < (1 << 16); i++)
			codes[i] = (uint8_t)((b->words[(o + i) >> 4] >> (30 - 2 * ((o + i) & 15))) & 3u);
		arcs_add_read(st->G, st->A, codes, (int)len, ord_base + r * ord_stride);
	}
	free(codes);
	return 0;
}

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double t_last;
static void phase(const char *name) { double t = now_ms(); fprintf(stderr, "[graphcheck] %-24s %9.1f ms\n", name, t - t_last); t_last = t; }

#define RD(ptr, sz, cnt) do { if (fread(ptr, sz, cnt, fi) != (size_t)(cnt)) { fprintf(stderr, "short dump\n"); return 2; } } while (0)

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: sdt-graphcheck <dump> <prefix>\n"); return 2; }
	FILE *fi = fopen(argv[1], "rb");
	if (!fi) return 2;
	int32_t hdr[6];
	uint64_t n;
	RD(hdr, 4, 6);
	RD(&n, 8, 1);
	const int K = hdr[0], nwv = hdr[1], nwk = hdr[2], p = hdr[3], d = hdr[4], dd = hdr[5];
	uint64_t *keys = (uint64_t *)malloc((n + 1) * (size_t)nwk * 8), *first = (uint64_t *)malloc((n + 1) * 8);
	uint32_t *ll = (uint32_t *)malloc((n + 1) * 4), *rf = (uint32_t *)malloc((n + 1) * 4), *cnt = (uint32_t *)malloc((n + 1) * 4);
	RD(keys, 8, n * nwk); RD(ll, 4, n); RD(rf, 4, n); RD(cnt, 4, n); RD(first, 8, n);
	fclose(fi);
	t_last = now_ms();
	if (sdt_test_env("SDT_GRAPHCHECK_A")) graph_init_kmerset_size = atoi(sdt_test_env("SDT_GRAPHCHECK_A"));      /* -a of the CLI */
	graph_t *G = graph_build(K, nwv, nwk, p, n, keys, ll, rf, cnt, first);
	phase("build");
	if (sdt_test_env("SDT_GRAPHCHECK_EMULATE")) graph_emulate_device(G);       /* the device-path commits on host-made records */
	graph_remove_minor_out(G, dd);
	phase("minor-out");
	if (!d) graph_remove_single_tips(G);
	phase("single tips");
	graph_remove_minor_tips(G);
	phase("minor tips");
	uint64_t ne = graph_build_edges(G, argv[2]);
	graph_edges_join(G);
	phase("edges");
	if (argc > 3) {
		sdt_cfg cfg;
		if (sdt_cfg_load(argv[3], &cfg) != 0) return 2;
		arc_state as = {G, arcs_new()};
		if (sdt_stream_reads(&cfg, cfg.max_rd_len ? cfg.max_rd_len : 100, 3, 50000, 0, arc_batch, &as, NULL) != 0) return 2;
		arcs_write(as.A, argv[2]);
		arcs_free(as.A);
		sdt_cfg_free(&cfg);
	}
	uint64_t nv = graph_write_vertex(G, argv[2]);
	graph_write_basic(argv[2], nv, K, ne, 0);
	graph_free(G);
	return 0;
}
