/* par.h -- tiny fork/join over an index range with dynamic chunks (pthreads) */
#ifndef SDT_PAR_H
#define SDT_PAR_H
#include <pthread.h>
#include <stdint.h>
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

static inline int par_threads(void)
{
	long n = sysconf(_SC_NPROCESSORS_ONLN);
	return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
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
