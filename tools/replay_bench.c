/* replay_bench.c -- host only: the layout replay of graph.c (put_kmerset / encap_kmerset emulation) on random keys, one set per
 * thread, to tune its prefetching on the target box.
 *   gcc -O2 -pthread [-DRP_AHEAD=32 -DRP_SECOND '-DRP_PF(p)=__builtin_prefetch((p),0,2)'] -o replay_bench tools/replay_bench.c -lm
 *   ./replay_bench <threads> <keys per set> */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <math.h>
#include <pthread.h>
#include <sys/mman.h>
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void *huge_malloc(size_t b) { void *p = NULL; const size_t h = 2u << 20, sz = (b + h - 1) & ~(h - 1); if (posix_memalign(&p, h, sz)) return NULL; if (!getenv("NO_THP")) madvise(p, sz, MADV_HUGEPAGE); return p; }
#define malloc(x) huge_malloc(x)
int graph_init_kmerset_size = 0;
static int prime_kh(uint64_t num) { if (num < 4) return 1; if (num % 2 == 0) return 0; uint64_t lim = (uint64_t)sqrt((float)num); for (uint64_t i = 3; i < lim; i += 2) if (num % i == 0) return 0; return 1; }
static uint64_t next_prime_kh(uint64_t n) { if (n % 2 == 0) n++; while (!prime_kh(n)) n += 2; return n; }
#include "replay_bench_inc.h"
typedef struct { uint64_t m, seed, cs; double ms; } job_t;
static void *run(void *v)
{
	job_t *J = (job_t *)v;
	uint64_t *keys = (uint64_t *)malloc(J->m * 8), *out = (uint64_t *)malloc(J->m * 8);
	uint64_t x = 88172645463325252ULL ^ (J->seed * 0x9E3779B97F4A7C15ULL);
	for (uint64_t i = 0; i < J->m; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; keys[i] = x >> 2; }
	const double t0 = now();
	replay_set1(keys, J->m, next_prime_kh(1024), 0, out);
	J->ms = now() - t0;
	uint64_t cs = 0;
	for (uint64_t i = 0; i < J->m; i++) cs = cs * 1000003 + out[i];
	J->cs = cs;
	return NULL;
}
#include <sys/syscall.h>
#include <unistd.h>
int main(int argc, char **argv)
{
	if (getenv("INTERLEAVE")) {                                /* MPOL_INTERLEAVE over the first N nodes */
		unsigned long mask[16] = {0};
		const int nodes = atoi(getenv("INTERLEAVE"));
		for (int i = 0; i < nodes; i++) mask[i / 64] |= 1UL << (i % 64);
		if (syscall(SYS_set_mempolicy, 3, mask, 1024) != 0) perror("set_mempolicy");
	}
	const int nt = argc > 1 ? atoi(argv[1]) : 16;
	const uint64_t m = argc > 2 ? strtoull(argv[2], 0, 10) : 42000000;
	pthread_t th[256];
	job_t J[256];
	const double t0 = now();
	for (int t = 0; t < nt; t++) { J[t].m = m; J[t].seed = (uint64_t)t; pthread_create(&th[t], NULL, run, &J[t]); }
	double worst = 0;
	uint64_t cs = 0;
	for (int t = 0; t < nt; t++) { pthread_join(th[t], NULL); if (J[t].ms > worst) worst = J[t].ms; cs ^= J[t].cs; }
	printf("%d sets x %llu keys: slowest replay %.0f ms = %.1f ns/key, wall %.0f ms, checksum %llx\n", nt, (unsigned long long)m, worst, worst * 1e6 / m, now() - t0, (unsigned long long)cs);
	return 0;
}
