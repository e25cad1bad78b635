/* graphcheck.c -- host-logic self test (no GPU): run the host graph phases of sdt-pregraph on a node dump.
 * dump (little endian): int32 K, nw_variant, nw_keys, p, d, dd; uint64 n; then keys[n*nw_keys] u64,
 * l_links[n] u32, r_flags[n] u32, count[n] u32, first[n] u64 -- exactly what sdt_gpu_export_nodes returns.
 * usage: sdt-graphcheck <dump> <out prefix> */
#include <stdio.h>
#include <stdlib.h>
#include "graph.h"

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
	graph_t *G = graph_build(K, nwv, nwk, p, n, keys, ll, rf, cnt, first);
	graph_remove_minor_out(G, dd);
	if (!d) graph_remove_single_tips(G);
	graph_remove_minor_tips(G);
	uint64_t ne = graph_build_edges(G, argv[2]);
	uint64_t nv = graph_write_vertex(G, argv[2]);
	graph_write_basic(argv[2], nv, K, ne, 0);
	graph_free(G);
	return 0;
}
