/* par.h -- tiny fork/join over an index range with dynamic chunks (pthreads) */
#ifndef SDT_PAR_H
#define SDT_PAR_H
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

typedef void (*par_body)(void *ctx, uint64_t lo, uint64_t hi, int tid);
typedef struct { par_body fn; void *ctx; uint64_t lo, hi, chunk; volatile uint64_t next; } par_job;
typedef struct { par_job *J; int tid; } par_arg;

static void *par_thread(void *a)
{
	par_arg *pa = (par_arg *)a;
	par_job *J = pa->J;
	for (;;) {
		const uint64_t b = __sync_fetch_and_add(&J->next, J->chunk);
		if (b >= J->hi) break;
		J->fn(J->ctx, b, b + J->chunk < J->hi ? b + J->chunk : J->hi, pa->tid);
	}
	return NULL;
}

/* usable CPUs: online CPUs, capped by the cgroup v2 CPU quota when there is one (more runnable threads than
 * quota only buys CFS throttling) */
static inline int par_threads(void)
{
	static int cached;
	if (cached) return cached;
	long n = sysconf(_SC_NPROCESSORS_ONLN);
	if (n < 1) n = 1;
	FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
	if (f) {
		long long quota = -1, period = 0;
		char q[64];
		if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && q[0] != 'm') {
			quota = atoll(q);
			if (quota > 0 && (quota + period - 1) / period < n) n = (long)((quota + period - 1) / period);
		}
		fclose(f);
	}
	if (n > 64) n = 64;
	cached = (int)n;
	return cached;
}

static inline void par_for(uint64_t lo, uint64_t hi, uint64_t chunk, par_body fn, void *ctx)
{
	if (hi <= lo) return;
	par_job J = {fn, ctx, lo, hi, chunk ? chunk : 4096, lo};
	int nt = par_threads();
	if ((hi - lo) / J.chunk + 1 < (uint64_t)nt) nt = (int)((hi - lo) / J.chunk + 1);
	pthread_t th[64];
	par_arg args[64];
	for (int t = 1; t < nt; t++) { args[t].J = &J; args[t].tid = t; pthread_create(&th[t], NULL, par_thread, &args[t]); }
	par_arg a0 = {&J, 0};
	par_thread(&a0);
	for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
}
#endif
