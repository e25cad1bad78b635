/* big.h -- the host graph lives in a handful of multi-gigabyte arrays that are accessed at random (nodes, look-up
 * index, replay tables): ask for transparent huge pages for every large block (the boxes run THP in `madvise` mode),
 * which cuts page faults, munmap time and, above all, TLB misses.  Small requests go to malloc unchanged; blocks from
 * here are released with free(). */
#ifndef SDT_BIG_H
#define SDT_BIG_H
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include "par.h"

#define SDT_BIG_MIN ((size_t)32 << 20)

static inline void *big_malloc(size_t bytes)
{
	if (bytes < SDT_BIG_MIN) return malloc(bytes);
	const size_t huge = (size_t)2 << 20, sz = (bytes + huge - 1) & ~(huge - 1);
	void *p = NULL;
	if (posix_memalign(&p, huge, sz) != 0) return NULL;
	static int thp = -1;                                /* SDT_NO_THP=1: plain pages (some sandboxes fault huge pages in very slowly) */
	if (thp < 0) thp = getenv("SDT_NO_THP") == NULL;
	if (thp) madvise(p, sz, MADV_HUGEPAGE);
	return p;
}

static void big_zero_part(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	memset((char *)vc + lo, 0, (size_t)(hi - lo));
}

static inline void *big_calloc(size_t n, size_t each)
{
	const size_t bytes = n * each;
	if (bytes < SDT_BIG_MIN) return calloc(n, each);
	void *p = big_malloc(bytes);
	if (p) par_for(0, bytes, (uint64_t)8 << 20, big_zero_part, p);
	return p;
}

#define malloc(x) big_malloc(x)
#define calloc(a, b) big_calloc(a, b)
#endif
