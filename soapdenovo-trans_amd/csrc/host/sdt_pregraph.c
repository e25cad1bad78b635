/* sdt_pregraph.c -- `sdt-pregraph -s configFile -o outputGraph [-K kmer -p n_cpu -d kmerFreqCutoff]`
 *
 * The reference's `pregraph` command line (pregraph.c:118-204) over the MI355X hashing path
 * (include/sdt_gpu.h).  Same options, same K clamp (pregraph.c:38-59), same stdout phrases for the
 * counters, and <prefix>.kmerFreq byte-identical to the reference (freqStat, prlHashReads.c:994-1023).
 * -p is the number of host parser threads here (the reference's worker pool is the GPU now).
 * There is no CPU fallback: without a gfx950 device the program exits non-zero.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <signal.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/wait.h>

#include "../sdt_knobs.h"
#include "../../../include/sdt_gpu.h"
#include "libcfg.h"
#include "seqio.h"
#include "readstream.h"
#include "graph/graph.h"
#include "graph/par.h"
#include "graph/big.h"

#ifndef SDT_MAX_K
#define SDT_MAX_K 127        /* one binary covers the 31/63/127mer variants; --max-k emulates a smaller one */
#endif

static void usage(int max_k)
{
	printf("\npregraph -s configFile -o outputGraph [-K kmer -p n_cpu -d kmerFreqCutoff]\n");
	printf("  -s\t<string>\tconfigFile: the config file of reads\n");
	printf("  -o\t<string>\toutputGraph: prefix of output graph file name\n");
	printf("  -K\t<int>\t\tkmer(min 13, max %d): kmer size, [23]\n", max_k);
	printf("  -p\t<int>\t\tn_cpu: number of cpu for use, [8]\n");
	printf("  -d\t<int>\t\tkmerFreqCutoff: kmers with frequency no larger than KmerFreqCutoff will be deleted, [0]\n");
}

typedef struct {
	sdt_ctx *gpu;
	unsigned long long reads;
	/* --gpus N: chunk i of the stream is counted by rank i % N; one collective push per group of N chunks */
	int rank, nranks, keep_all, keep_mine, fill, have;
	unsigned long long kept_reads;     /* reads handed to sdt_gpu_keep_reads (a rank that owns no chunk of the input keeps none) */
	int K, hinted;
	uint64_t total_text;               /* bytes of all input files (0: unknown) */
	uint32_t *w;
	uint64_t *o, nw, n, cap_w, cap_o, ord_base, ord_stride;
	/* --gpus N without a rank that keeps every read: nobody scans foreign chunks (seqio.h: sdt_read_shard_skip_foreign).  The record
	 * counts of a group's chunks are gathered when the group is complete, and only then does a rank know the ordinals of its own */
	int defer;
	struct { int sid, stride, parity; uint64_t n; } grp[64];
	sdt_stream_ordinals ords;
} push_state;

/* millisecond phase timer on stderr (the reference's own lines on stdout have 1 s resolution) */
static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static double g_t_last, g_t_main;
static int g_quiet;                 /* --gpus N: ranks > 0 */
static void phase(const char *name)
{
	if (g_quiet) return;
	const double t = now_ms();
	fprintf(stderr, "[sdt-pregraph] %-28s %9.1f ms\n", name, t - g_t_last);
	g_t_last = t;
}

/* bytes of every file pass 1 will read (readstream.c's selection: asm_flags 1 or 3) */
static uint64_t input_bytes(const sdt_cfg *cfg)
{
	uint64_t total = 0;
	for (int i = 0; i < cfg->nlibs; i++) {
		const sdt_lib *l = &cfg->libs[i];
		if (l->asm_flag != 1 && l->asm_flag != 3) continue;
		char **lists[] = {l->f1, l->f2, l->q1, l->q2, l->p, l->f, l->q};
		const int counts[] = {l->nf1, l->nf2, l->nq1, l->nq2, l->np, l->nf, l->nq};
		for (int k = 0; k < 7; k++)
			for (int j = 0; j < counts[k]; j++) {
				struct stat sb;
				if (stat(lists[k][j], &sb) == 0) total += (uint64_t)sb.st_size;
			}
	}
	return total;
}

/* pooled batches (pinned buffers, seqio.h) are pushed asynchronously: the buffer goes back to the pool once its copy has left it,
 * a few pushes later */
#define PUSH_DEPTH 12
static struct { int slot[PUSH_DEPTH]; uint64_t ticket[PUSH_DEPTH]; int head, n; } g_inflight;

static int inflight_retire(sdt_ctx *gpu, int down_to)
{
	while (g_inflight.n > down_to) {
		if (sdt_gpu_push_wait(gpu, g_inflight.ticket[g_inflight.head]) != SDT_OK) { fprintf(stderr, "sdt_gpu_push_wait: %s\n", sdt_gpu_last_error()); return -1; }
		sdt_pool_release(g_inflight.slot[g_inflight.head]);
		g_inflight.head = (g_inflight.head + 1) % PUSH_DEPTH;
		g_inflight.n--;
	}
	return 0;
}

static int push_batch(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride)
{
	push_state *st = (push_state *)user;
	unsigned long long before = st->reads / 1000000ULL;
	st->reads += b->nreads;
	if (st->reads / 1000000ULL != before)
		printf("--- %lluth reads\n", st->reads / 1000000ULL * 1000000ULL);    /* prlHashReads.c:587-588 */
	static int parse_only = -1;                              /* SDT_PARSE_ONLY=1 (measurement): parse and pack, push nothing */
	if (parse_only < 0) parse_only = sdt_tuning_env("SDT_PARSE_ONLY") != NULL;
	if (parse_only) return 0;
	if (st->total_text && !st->hinted && b->text_bytes && b->nreads) {
		/* the size of the job from its first chunk: k-mers per byte of text x bytes of all files (the pools of the locality
		 * pipeline are then the job's from the first batch on) */
		uint64_t km = 0;
		for (uint64_t i = 0; i < b->nreads; i++) { const uint64_t len = b->offsets[i + 1] - b->offsets[i]; if (len > (uint64_t)st->K) km += len - (uint64_t)st->K + 1; }
		sdt_gpu_hint_total_kmers(st->gpu, (uint64_t)((double)km / (double)b->text_bytes * (double)st->total_text));
		st->hinted = 1;
	}
	sdt_gpu_set_read_ordinal(st->gpu, ord_base, ord_stride);
	if (b->pool_slot >= 0) {
		uint64_t ticket = 0;
		const int rc = b->fixed_len ? sdt_gpu_push_reads_fixed_async(st->gpu, b->words, b->nwords, b->nreads, b->fixed_len, &ticket)
		                            : sdt_gpu_push_reads_async(st->gpu, b->words, b->nwords, b->offsets, b->nreads, &ticket);
		if (rc != SDT_OK) { fprintf(stderr, "sdt_gpu_push_reads_async: %s\n", sdt_gpu_last_error()); return -1; }
		if (inflight_retire(st->gpu, PUSH_DEPTH - 1) != 0) return -1;
		const int at = (g_inflight.head + g_inflight.n) % PUSH_DEPTH;
		g_inflight.slot[at] = b->pool_slot;
		g_inflight.ticket[at] = ticket;
		g_inflight.n++;
		sdt_pool_take(b->pool_slot);
		return 0;
	}
	if (sdt_gpu_push_reads(st->gpu, b->words, b->nwords, b->offsets, b->nreads) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_push_reads: %s\n", sdt_gpu_last_error());
		return -1;
	}
	return 0;
}

static void count_reads_line(push_state *st, uint64_t n)
{
	const unsigned long long before = st->reads / 1000000ULL;
	st->reads += n;
	if (st->reads / 1000000ULL != before)
		printf("--- %lluth reads\n", st->reads / 1000000ULL * 1000000ULL);    /* prlHashReads.c:587-588 */
}

static int flush_group(push_state *st)
{
	static const uint32_t none[4] = {0, 0, 0, 0};
	static const uint64_t zero[1] = {0};
	if (st->defer) {
		/* who parsed what: one all-gather (a sum over vectors with one non-zero entry each) per group of nranks chunks, then every rank
		 * does the arithmetic of readstream.c for the group's chunks in order */
		int64_t cnt[64];
		memset(cnt, 0, sizeof cnt);
		if (st->have) cnt[st->rank] = (int64_t)st->grp[st->rank].n;
		if (sdt_gpu_allreduce_i64(st->gpu, cnt, st->nranks) != SDT_OK) { fprintf(stderr, "[rank %d] record counts of a group: %s\n", st->rank, sdt_gpu_last_error()); return -1; }
		for (int i = 0; i < st->fill; i++) {
			const uint64_t base = sdt_stream_ordinals_next(&st->ords, st->grp[i].sid, st->grp[i].stride, st->grp[i].parity, (uint64_t)cnt[i]);
			if (i == st->rank) { st->ord_base = base; st->ord_stride = (uint64_t)st->grp[i].stride; }
			count_reads_line(st, (uint64_t)cnt[i]);
		}
		if (st->have && st->keep_mine && st->n) {
			sdt_gpu_set_read_ordinal(st->gpu, st->ord_base, st->ord_stride);
			if (sdt_gpu_keep_reads(st->gpu, st->w, st->nw, st->o, st->n) != SDT_OK) { fprintf(stderr, "sdt_gpu_keep_reads: %s\n", sdt_gpu_last_error()); return -1; }
			st->kept_reads += st->n;
		}
	}
	if (st->have) sdt_gpu_set_read_ordinal(st->gpu, st->ord_base, st->ord_stride);
	const int rc = sdt_gpu_push_reads_sharded(st->gpu, st->have ? st->w : none, st->have ? st->nw : 4, st->have ? st->o : zero, st->have ? st->n : 0);
	if (rc != SDT_OK) fprintf(stderr, "[rank %d] sdt_gpu_push_reads_sharded: %s\n", st->rank, sdt_gpu_last_error());
	st->have = 0;
	st->fill = 0;
	return rc == SDT_OK ? 0 : -1;
}

static int push_batch_sharded(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride)
{
	push_state *st = (push_state *)user;
	if (st->defer) {
		if (b->owner < 0 || b->owner >= 64 || b->owner != st->fill) { fprintf(stderr, "[rank %d] chunk %llu out of turn\n", st->rank, (unsigned long long)b->chunk_index); return -1; }
		st->grp[b->owner].sid = b->stream_id; st->grp[b->owner].stride = (int)ord_stride; st->grp[b->owner].parity = b->stream_parity;
		st->grp[b->owner].n = b->count_unknown ? 0 : b->nreads;
	} else
		count_reads_line(st, b->nreads);
	if (!st->defer && ((st->keep_all && !b->counted_only) || (st->keep_mine && b->owner == st->rank)) && b->nreads) {
		/* the reads of the second pass stay resident: this rank's own share (every rank maps its reads), or -- rank 0 without that -- all */
		sdt_gpu_set_read_ordinal(st->gpu, ord_base, ord_stride);
		if (sdt_gpu_keep_reads(st->gpu, b->words, b->nwords, b->offsets, b->nreads) != SDT_OK) {
			fprintf(stderr, "sdt_gpu_keep_reads: %s\n", sdt_gpu_last_error());
			return -1;
		}
		st->kept_reads += b->nreads;
	}
	if (b->owner == st->rank && b->nreads) {                      /* mine: it waits for the end of its group */
		if (b->nwords > st->cap_w) { st->cap_w = b->nwords * 5 / 4; st->w = (uint32_t *)realloc(st->w, st->cap_w * 4); }
		if (b->nreads + 1 > st->cap_o) { st->cap_o = (b->nreads + 1) * 5 / 4; st->o = (uint64_t *)realloc(st->o, st->cap_o * 8); }
		if (!st->w || !st->o) { fprintf(stderr, "[rank %d] out of host memory for a batch of %llu reads\n", st->rank, (unsigned long long)b->nreads); return -1; }
		memcpy(st->w, b->words, b->nwords * 4);
		memcpy(st->o, b->offsets, (b->nreads + 1) * 8);
		st->nw = b->nwords; st->n = b->nreads; st->ord_base = ord_base; st->ord_stride = ord_stride;
		st->have = 1;
	}
	if (++st->fill == st->nranks)
		return flush_group(st);
	return 0;
}

/* --gpus N: the shard of every other rank comes to rank 0 through POSIX shared memory */
typedef struct {
	volatile int ready; sdt_comm_id id; char name[64];
	/* second read pass on every rank (round 5): rank 0 publishes the graph (key -> path word, patch table) in shared memory and says
	 * so here; every rank maps ITS reads and leaves its arcs in a segment of its own */
	volatile int paths_state;                  /* 0 not yet, 1 published, -1 no per-rank pass (the ranks may go), 2 rank 0 has all arcs */
	volatile unsigned long long paths_n, patch_n, num_ed;
	volatile int arcs_state[64];               /* 1: this rank's arcs are in its segment, -1: it failed */
	volatile unsigned long long arcs_n[64], arcs_reads[64];
} boot_t;

/* Failure propagation between the forked ranks.  RCCL collectives have no timeout: a rank that leaves early would keep its
 * peers blocked for ever, holding their GPUs.  So: every child dies with its parent (PR_SET_PDEATHSIG); rank 0 -- the parent --
 * reaps children in a SIGCHLD handler and, when one of them failed, kills the rest and exits; and whenever rank 0 itself
 * leaves (any `return`, atexit) it takes the children that are still alive with it. */
static volatile pid_t g_child[64];
static volatile int g_nchild;
static char g_shm_name[64];                 /* rank 0: the job's name in /dev/shm ("" = no segments of ours) */
static int g_shm_ranks;

/* every segment the job may have made, whoever made it (a rank killed while polling never unlinks its own): rank 0 calls this
 * on every way out -- unlinking a name that is not there costs a failed system call */
static void unlink_segments(void)
{
	char seg[128];
	if (!g_shm_name[0]) return;
	snprintf(seg, sizeof seg, "/sdt_%s_paths", g_shm_name);
	shm_unlink(seg);
	for (int r = 1; r < g_shm_ranks; r++) {
		snprintf(seg, sizeof seg, "/sdt_%s_a%d", g_shm_name, r);
		shm_unlink(seg);
		snprintf(seg, sizeof seg, "/sdt_%s_n%d", g_shm_name, r);
		shm_unlink(seg);
	}
}

static void kill_children(void)
{
	for (int i = 0; i < g_nchild; i++)
		if (g_child[i] > 0) { kill(g_child[i], SIGKILL); (void)waitpid(g_child[i], NULL, 0); g_child[i] = 0; }
	unlink_segments();
}

static void on_sigchld(int sig)
{
	(void)sig;
	int st;
	/* only the ranks recorded in g_child are reaped here (another child of the process is not this handler's business) */
	for (int k = 0; k < g_nchild; k++) {
		const pid_t c = g_child[k];
		if (c <= 0 || waitpid(c, &st, WNOHANG) != c) continue;
		g_child[k] = 0;
		if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
			static const char msg[] = "sdt-pregraph: a rank failed; stopping the others\n";
			if (write(2, msg, sizeof msg - 1) < 0) { }
			for (int i = 0; i < g_nchild; i++)
				if (g_child[i] > 0) kill(g_child[i], SIGKILL);
			unlink_segments();
			_exit(1);
		}
	}
}

/* first touch of a fresh segment, on all threads: the pages of a new shared-memory object are made (and cleared) by the kernel at the
 * first write, one fault per page on the writing thread -- the device-to-host copy of a 2 GB path table into a fresh segment ran at
 * 3 GB/s (685 ms at 20 M reads with four ranks, profiles/r6) when its four staging threads took those faults one by one */
static void touch_pages(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	volatile char *b = (volatile char *)vc;
	for (uint64_t pg = lo; pg < hi; pg++) b[pg << 12] = 0;
}

static void *shm_region(const char *name, size_t bytes, int create)
{
	int fd = create ? shm_open(name, O_CREAT | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
	if (fd < 0) return NULL;
	if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return NULL; }
	void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) return NULL;
	if (create && bytes >= ((size_t)64 << 20)) par_for(0, (bytes + 4095) >> 12, 4096, touch_pages, p);
	return p;
}

/* arcs of several ranks in one array: equal (from, to) pairs become one arc -- multiplicities add up, the first occurrence is the
 * smallest.  Returns the number of arcs left (in the front of the arrays). */
typedef struct { uint64_t key, ord; uint32_t mult; } arc_rec;
static int arc_rec_cmp(const void *a, const void *b)
{
	const arc_rec *x = (const arc_rec *)a, *y = (const arc_rec *)b;
	return x->key < y->key ? -1 : (x->key > y->key ? 1 : (x->ord < y->ord ? -1 : (x->ord > y->ord)));
}
static uint64_t arcs_combine(uint32_t *from, uint32_t *to, uint32_t *mult, uint64_t *ord, uint64_t n)
{
	arc_rec *v = (arc_rec *)malloc((n + 1) * sizeof(arc_rec));
	if (!v) { fprintf(stderr, "out of host memory for %llu arcs\n", (unsigned long long)n); exit(1); }
	for (uint64_t i = 0; i < n; i++) { v[i].key = ((uint64_t)from[i] << 32) | to[i]; v[i].ord = ord[i]; v[i].mult = mult[i]; }
	qsort(v, n, sizeof(arc_rec), arc_rec_cmp);
	uint64_t m = 0;
	for (uint64_t i = 0; i < n;) {
		uint64_t j = i, sum = 0;
		while (j < n && v[j].key == v[i].key) sum += v[j++].mult;
		from[m] = (uint32_t)(v[i].key >> 32); to[m] = (uint32_t)v[i].key; ord[m] = v[i].ord;      /* (sorted: the smallest ordinal first) */
		mult[m] = sum > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)sum;
		m++;
		i = j;
	}
	free(v);
	return m;
}

/* second pass (prlRead2edge): unpack each read and thread it through the edge graph */
typedef struct {
	graph_t *G;
	struct arcs *A;
	unsigned long long reads;
} arc_state;

static int arc_batch(void *user, const sdt_batch *b, uint64_t ord_base, uint64_t ord_stride)
{
	arc_state *st = (arc_state *)user;
	uint8_t *codes = NULL;
	size_t cap = 0;
	for (uint64_t r = 0; r < b->nreads; r++) {
		const uint64_t o = b->offsets[r], len = b->offsets[r + 1] - o;
		if (len > cap) { cap = len * 2 + 64; codes = (uint8_t *)realloc(codes, cap); }
		for (uint64_t i = 0; i < len; i++)
			codes[i] = (uint8_t)((b->words[(o + i) >> 4] >> (30 - 2 * ((o + i) & 15))) & 3u);
		arcs_add_read(st->G, st->A, codes, (int)len, ord_base + r * ord_stride);
	}
	st->reads += b->nreads;
	free(codes);
	return 0;
}

/* graph.h dev_walks: bring the device mirror up to date with what the host wrote, then take the walks of every
 * node from it (sdt_gpu_tip_walks) */
typedef struct { sdt_ctx *gpu; int nwk, indexed, by_index; } dev_state;     /* by_index: the device numbered the nodes itself (sdt_gpu_layout_apply) */

static void gather_keys(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	void **a = (void **)vc;
	const graph_t *g = (const graph_t *)a[0];
	uint64_t *k = (uint64_t *)a[1];
	const int nwk = (int)(intptr_t)a[2];
	for (uint64_t i = lo; i < hi; i++)
		for (int w = 0; w < nwk; w++) k[i * nwk + w] = g->nodes[i].seq.w[4 - nwk + w];
}

/* what the second read pass needs of every node (sdt_gpu_load_paths): skip flag, linear, twin, edge id */
static void gather_paths(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	void **a = (void **)vc;
	const graph_t *G = (const graph_t *)a[0];
	uint64_t *pk = (uint64_t *)a[1], *pw = (uint64_t *)a[2];
	const int nwk = (int)(intptr_t)a[3];
	for (uint64_t i = lo; i < hi; i++) {
		const gnode_t *nd = &G->nodes[i];
		if (pk) for (int w = 0; w < nwk; w++) pk[i * nwk + w] = nd->seq.w[4 - nwk + w];
		const int skip = nd->deleted || (nd->linear && !nd->inEdge);
		pw[i] = (uint64_t)skip | ((uint64_t)nd->linear << 1) | ((uint64_t)nd->twin << 2) | ((uint64_t)nd->l_links << 32);
	}
}

static void gather_dirty(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	void **a = (void **)vc;
	const graph_t *g = (const graph_t *)a[0];
	uint64_t *k = (uint64_t *)a[1];
	uint32_t *l = (uint32_t *)a[2], *r = (uint32_t *)a[3];
	const int nwk = (int)(intptr_t)a[4];
	for (uint64_t j = lo; j < hi; j++) {
		const gnode_t *nd = &g->nodes[g->dlist[j]];
		if (k) for (int w = 0; w < nwk; w++) k[j * nwk + w] = nd->seq.w[4 - nwk + w];
		l[j] = nd->l_links;
		r[j] = nd->r_links | ((uint32_t)nd->linear << 24) | ((uint32_t)nd->deleted << 25);
	}
}

/* the device table mirrors the host graph: node order known to the device, nodes written since the last call sent over */
static int dev_mirror_sync_(graph_t *g);
static int dev_mirror_sync(graph_t *g)
{
	const double t0 = now_ms();
	const size_t dn = g->dn;
	const int rc = dev_mirror_sync_(g);
	if (sdt_env("SDT_TIMING") && !g_quiet) fprintf(stderr, "[device]   mirror sync: %zu nodes written by the host sent over in %.1f ms\n", dn, now_ms() - t0);
	return rc;
}
static int dev_mirror_sync_(graph_t *g)
{
	dev_state *D = (dev_state *)g->dev_user;
	const int nwk = D->nwk;
	if (!D->indexed) {
		uint64_t *k = (uint64_t *)malloc((g->n + 1) * (size_t)nwk * 8);
		void *ga[3] = {g, k, (void *)(intptr_t)nwk};
		par_for(0, g->n, 1 << 16, gather_keys, ga);
		const int rc = sdt_gpu_set_node_index(D->gpu, k, g->n);
		free(k);
		if (rc != SDT_OK) { fprintf(stderr, "sdt_gpu_set_node_index: %s\n", sdt_gpu_last_error()); return 1; }
		D->indexed = 1;
	}
	if (g->dn && D->by_index) {                                /* what the host wrote goes over by node index: no keys, no look-ups */
		uint32_t *l = (uint32_t *)malloc(g->dn * 4), *r = (uint32_t *)malloc(g->dn * 4);
		void *da[5] = {g, NULL, l, r, (void *)(intptr_t)nwk};
		par_for(0, g->dn, 1 << 14, gather_dirty, da);
		const int rc = sdt_gpu_update_nodes_by_index(D->gpu, g->dlist, l, r, g->dn);
		free(l); free(r);
		if (rc != SDT_OK) { fprintf(stderr, "sdt_gpu_update_nodes_by_index: %s\n", sdt_gpu_last_error()); return 1; }
	} else if (g->dn) {
		uint64_t *k = (uint64_t *)malloc(g->dn * (size_t)nwk * 8);
		uint32_t *l = (uint32_t *)malloc(g->dn * 4), *r = (uint32_t *)malloc(g->dn * 4);
		void *da[5] = {g, k, l, r, (void *)(intptr_t)nwk};
		par_for(0, g->dn, 1 << 14, gather_dirty, da);
		const int rc = sdt_gpu_update_nodes(D->gpu, k, l, r, g->dn);
		free(k); free(l); free(r);
		if (rc != SDT_OK) { fprintf(stderr, "sdt_gpu_update_nodes: %s\n", sdt_gpu_last_error()); return 1; }
	}
	return 0;
}

/* graph_index_hook: the device knows every node's place in the host array, let it build the host's look-up index */
static int dev_index_hook(graph_t *g, void *user)
{
	dev_state *D = (dev_state *)user;
	g->dev_user = D;
	const double t0 = now_ms();
	if (dev_mirror_sync(g) != 0) exit(1);
	const double t1 = now_ms();
	if (g->index64)
		return 1;                                                        /* a graph past 2^32 nodes: the host builds its 64-bit index */
	uint64_t cap = 1024;
	while (cap < 2 * g->n + 2) cap <<= 1;
	g->index = (uint32_t *)malloc(cap * sizeof(uint32_t));
	g->index_mask = cap - 1;
	if (!g->index || sdt_gpu_build_host_index(D->gpu, g->index, cap) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_build_host_index: %s\n", sdt_gpu_last_error());
		exit(1);
	}
	if (sdt_env("SDT_TIMING")) fprintf(stderr, "[graph]      node order to the device %.1f ms, index built + copied back %.1f ms\n", t1 - t0, now_ms() - t1);
	return 0;
}

static int dev_edge_ports_hook(graph_t *g, uint64_t **records, uint64_t *nr)
{
	dev_state *D = (dev_state *)g->dev_user;
	if (dev_mirror_sync(g) != 0) return 1;
	uint64_t cap = g->n / 8 + 4096;
	for (;;) {
		uint64_t *rec = (uint64_t *)malloc(cap * 17 * sizeof(uint64_t));
		if (!rec) { fprintf(stderr, "out of memory for %llu port records\n", (unsigned long long)cap); return 1; }
		const int rc = sdt_gpu_edge_ports(D->gpu, rec, cap, nr);
		if (rc == SDT_OK) { *records = rec; return 0; }
		free(rec);
		if (rc == SDT_EFULL && *nr > cap) { cap = *nr; continue; }
		fprintf(stderr, "sdt_gpu_edge_ports: %s\n", sdt_gpu_last_error());
		return 1;
	}
}

static int dev_build_edges_hook(graph_t *g, uint64_t **records, int *key_words, uint64_t *n_edges, uint64_t *num_ed, char **bases, uint64_t *n_bases)
{
	dev_state *D = (dev_state *)g->dev_user;
	if (dev_mirror_sync(g) != 0) return 1;
	const int rc = sdt_gpu_build_edges(D->gpu, n_edges, num_ed, n_bases);
	if (rc == SDT_ESTATE && strstr(sdt_gpu_last_error(), "does not lead back")) return 2;       /* not symmetric: the sequential way */
	if (rc != SDT_OK) { fprintf(stderr, "sdt_gpu_build_edges: %s\n", sdt_gpu_last_error()); return 1; }
	const int rw = 4 + 2 * D->nwk;
	uint64_t *rec = (uint64_t *)malloc((*n_edges + 1) * (size_t)rw * sizeof(uint64_t));
	char *b = (char *)malloc(*n_bases + 16);
	if (!rec || !b) { fprintf(stderr, "out of memory for %llu edges\n", (unsigned long long)*n_edges); return 1; }
	if (sdt_gpu_fetch_records(D->gpu, rec, *n_edges * (uint64_t)rw) != SDT_OK || sdt_gpu_fetch_edge_bases(D->gpu, b, *n_bases) != SDT_OK) {
		fprintf(stderr, "edge records: %s\n", sdt_gpu_last_error());
		return 1;
	}
	*records = rec;
	*bases = b;
	*key_words = D->nwk;
	return 0;
}

static int dev_minor_out_hook(graph_t *g, double threshold, uint64_t **records, uint64_t *nj, uint64_t *nr)
{
	dev_state *D = (dev_state *)g->dev_user;
	if (dev_mirror_sync(g) != 0) return 1;
	const double t0 = now_ms();
	if (sdt_gpu_minor_out_labelled(D->gpu, threshold, nj, nr) != SDT_OK) { fprintf(stderr, "sdt_gpu_minor_out_labelled: %s\n", sdt_gpu_last_error()); return 1; }
	const double t1 = now_ms();
	uint64_t *rec = (uint64_t *)malloc((*nr + 1) * MO_RW * sizeof(uint64_t));
	if (!rec) { fprintf(stderr, "out of memory for %llu junction records\n", (unsigned long long)*nr); return 1; }
	if (sdt_gpu_fetch_records(D->gpu, rec, *nr * MO_RW) != SDT_OK) { fprintf(stderr, "sdt_gpu_fetch_records: %s\n", sdt_gpu_last_error()); free(rec); return 1; }
	if (sdt_env("SDT_TIMING") && !g_quiet) fprintf(stderr, "[device]   junction dry run + components %.1f ms, %llu + %llu records fetched in %.1f ms\n", t1 - t0, (unsigned long long)*nj, (unsigned long long)(*nr - *nj), now_ms() - t1);
	*records = rec;
	return 0;
}

/* removeMinorOut's commit on the device, the long components on the host's threads at the same time (graph.h) */
static int dev_minor_out_commit_begin_hook(graph_t *g, double threshold, uint64_t **skipped, uint64_t *n_skipped, uint64_t *n_skipped_records)
{
	dev_state *D = (dev_state *)g->dev_user;
	if (dev_mirror_sync(g) != 0) return 1;
	const double t0 = now_ms();
	uint64_t nj = 0, nr = 0, largest = 0, nsk = 0, nskr = 0;
	if (sdt_gpu_minor_out_labelled(D->gpu, threshold, &nj, &nr) != SDT_OK) { fprintf(stderr, "sdt_gpu_minor_out_labelled: %s\n", sdt_gpu_last_error()); return 1; }
	const double t1 = now_ms();
	/* one lane walks a component at about a microsecond per dependent access (~100 us per visit); a host thread takes ~150 ns */
	/* (the device's lanes and the host's threads work side by side: at 200 M reads 1536 / 2048 / 2560 / 3072 / 4096 visits per component gave
	 * 983 / 895 / 829 / 813 / 1020 ms for the whole pass -- past 3072 the host waits for the longest lane, profiles/r5/README.md) */
	/* Round 6: the limit follows the size of the job.  The device's part lasts as long as its longest lane -- limit x ~50 us, whatever the
	 * job --, the host's part grows with the visits of the components past the limit: 3072 is where they meet at 16.5 M visits (200 M
	 * reads); at 1.8 M visits (8 M reads) the host was done with its six long components after 7 ms and then waited 154 ms for the lanes of
	 * 3072 visits (profiles/r6/e2e_8M_se_ours_only.json).  visits / 5000, within 256 .. 3072. */
	uint64_t by_size = nj / 5000;
	if (by_size < 256) by_size = 256;
	if (by_size > 3072) by_size = 3072;
	const uint64_t max_comp = sdt_test_env("SDT_COMMIT_MAX_COMPONENT") ? strtoull(sdt_test_env("SDT_COMMIT_MAX_COMPONENT"), NULL, 10) : by_size;
	if (sdt_gpu_minor_out_commit_begin(D->gpu, threshold, max_comp, &largest, &nsk, &nskr) != SDT_OK) { fprintf(stderr, "sdt_gpu_minor_out_commit_begin: %s\n", sdt_gpu_last_error()); return 1; }
	const double t2 = now_ms();
	uint64_t *sk = (uint64_t *)malloc((nskr + 1) * MO_RW * sizeof(uint64_t));
	if (!sk) { fprintf(stderr, "out of memory for %llu records\n", (unsigned long long)nskr); return 1; }
	if (sdt_gpu_fetch_skipped(D->gpu, sk, nskr) != SDT_OK) { fprintf(stderr, "sdt_gpu_fetch_skipped: %s\n", sdt_gpu_last_error()); return 1; }
	if (nsk) *skipped = sk; else { free(sk); *skipped = NULL; }
	*n_skipped = nsk;
	*n_skipped_records = nskr;
	if (sdt_env("SDT_TIMING") && !g_quiet)
		fprintf(stderr, "[device]   junction dry run + components %.1f ms (%llu visits, largest component %llu), components + long ones gathered %.1f ms, %llu records of long components fetched in %.1f ms\n",
		        t1 - t0, (unsigned long long)nj, (unsigned long long)largest, t2 - t1, (unsigned long long)nskr, now_ms() - t2);
	return 0;
}

static int dev_minor_out_commit_finish_hook(graph_t *g, uint64_t *off, uint64_t *linear)
{
	dev_state *D = (dev_state *)g->dev_user;
	const double t0 = now_ms();
	uint64_t nw = 0;
	if (sdt_gpu_minor_out_commit_finish(D->gpu, off, linear, &nw) != SDT_OK) { fprintf(stderr, "sdt_gpu_minor_out_commit_finish: %s\n", sdt_gpu_last_error()); return 1; }
	const double t1 = now_ms();
	uint64_t *node = (uint64_t *)malloc((nw + 1) * sizeof(uint64_t));
	uint32_t *l = (uint32_t *)malloc((nw + 1) * sizeof(uint32_t)), *r = (uint32_t *)malloc((nw + 1) * sizeof(uint32_t));
	if (!node || !l || !r) { fprintf(stderr, "out of memory for %llu written nodes\n", (unsigned long long)nw); return 1; }
	if (sdt_gpu_fetch_written(D->gpu, node, l, r, nw) != SDT_OK) { fprintf(stderr, "sdt_gpu_fetch_written: %s\n", sdt_gpu_last_error()); return 1; }
	const double t2 = now_ms();
	graph_apply_written(g, node, l, r, nw);
	free(node); free(l); free(r);
	if (sdt_env("SDT_TIMING") && !g_quiet)
		fprintf(stderr, "[device]   waited %.1f ms for the device's visits + marking + list, %llu written nodes fetched in %.1f ms, applied in %.1f ms\n",
		        t1 - t0, (unsigned long long)nw, t2 - t1, now_ms() - t2);
	return 0;
}

static int dev_walks_hook(graph_t *g, int thin, int cut_len, uint64_t **records, uint64_t *nr)
{
	dev_state *D = (dev_state *)g->dev_user;
	if (dev_mirror_sync(g) != 0) return 1;
	const double t0 = now_ms();
	if (sdt_gpu_tip_walks_labelled(D->gpu, thin, cut_len, nr) != SDT_OK) { fprintf(stderr, "sdt_gpu_tip_walks_labelled: %s\n", sdt_gpu_last_error()); return 1; }
	const double t1 = now_ms();
	uint64_t *rec = (uint64_t *)malloc((*nr + 1) * 3 * sizeof(uint64_t));
	if (!rec) { fprintf(stderr, "out of memory for %llu walk records\n", (unsigned long long)*nr); return 1; }
	if (sdt_gpu_fetch_records(D->gpu, rec, *nr * 3) != SDT_OK) { fprintf(stderr, "sdt_gpu_fetch_records: %s\n", sdt_gpu_last_error()); free(rec); return 1; }
	if (sdt_env("SDT_TIMING") && !g_quiet) fprintf(stderr, "[device]   walks + components %.1f ms, %llu records fetched in %.1f ms\n", t1 - t0, (unsigned long long)*nr, now_ms() - t1);
	*records = rec;
	return 0;
}

/* *.vertex written beside the second read pass (the device is busy, the host is not), then the node array is let go */
typedef struct { graph_t *G; const char *prefix; uint64_t nv; } vx_job;
static void *vertex_thread(void *v)
{
	vx_job *J = (vx_job *)v;
	J->nv = graph_write_vertex(J->G, J->prefix);
	graph_free_later(J->G->nodes, J->G->index, J->G->dirty, J->G->dlist);
	J->G->nodes = NULL; J->G->index = NULL; J->G->dirty = NULL; J->G->dlist = NULL;
	return NULL;
}

int main(int argc, char **argv)
{
	char cfgfile[4096] = "", prefix[4096] = "";
	int K = 23, threads = 8, d = 0, max_k = 0, device = 0, dd = 5, hash_only = 0, host_map = 0, host_walks = 0;
	int gpus = 1, share_device = 0, rank = 0;
	int have_s = 0, have_o = 0, c;
	unsigned long long est = 0;
	static struct option longopts[] = {{"max-k", required_argument, 0, 1000}, {"device", required_argument, 0, 1001},
	                                   {"est-distinct", required_argument, 0, 1002}, {"hash-only", no_argument, 0, 1003}, {"host-map", no_argument, 0, 1004}, {"host-walks", no_argument, 0, 1005},
	                                   {"gpus", required_argument, 0, 1006}, {"share-device", no_argument, 0, 1007},
	                                   {0, 0, 0, 0}};
	/* accept an optional leading "pregraph" sub-command like the reference's dispatcher (main.c:49-106) */
	if (argc > 1 && strcmp(argv[1], "pregraph") == 0) { argv++; argc--; }
	while ((c = getopt_long(argc, argv, "a:s:o:K:p:d:Di:n", longopts, NULL)) != -1) {
		switch (c) {
		case 's': have_s = 1; snprintf(cfgfile, sizeof cfgfile, "%s", optarg); break;
		case 'o': have_o = 1; snprintf(prefix, sizeof prefix, "%s", optarg); break;
		case 'K': K = atoi(optarg); break;
		case 'p': threads = atoi(optarg); break;
		case 'd': d = atoi(optarg) >= 0 ? atoi(optarg) : 0; break;          /* pregraph.c:159 */
		case 'i': dd = atoi(optarg) >= 0 ? atoi(optarg) : 0; break;          /* pregraph.c:170-173 */
		case 'a': graph_init_kmerset_size = atoi(optarg); break;             /* pregraph.c:160-162; layout replay, graph/graph.c */
		case 'D': break;                                                     /* accepted, commented out upstream (pregraph.c:155-158) */
		case 'n':
			fprintf(stderr, "-n (N-aware k-mers) is not supported: the reference path is broken (survey 9.3-q11)\n");
			return 1;
		case 1000: max_k = atoi(optarg); break;
		case 1001: device = atoi(optarg); break;
		case 1002: est = strtoull(optarg, NULL, 10); break;
		case 1003: hash_only = 1; break;
		case 1004: host_map = 1; break;
		case 1005: host_walks = 1; break;
		case 1006: gpus = atoi(optarg); break;                /* one process per GPU: devices --device .. --device + N - 1 */
		case 1007: share_device = 1; break;                   /* validation: all ranks on --device, shared-memory transport */
		default:
			if (!have_s || !have_o) { usage(max_k ? max_k : SDT_MAX_K); return 255; }
		}
	}
	/* which reference binary is being stood in for: it fixes the words per printed k-mer and the bytes hash_kmer
	 * runs over (31mer / 63mer / 127mer); default = the smallest shipped variant that can hold K */
	if (max_k == 0) max_k = K <= 31 ? 31 : SDT_MAX_K;
	if (!have_s || !have_o) { usage(max_k); return 255; }
	if (d > 127) d = (signed char)d;                                        /* deLowKmer is a char (survey q12) */
	/* pregraph.c:38-59 */
	if (K % 2 == 0) { K++; printf("K should be an odd number\n"); }
	if (K < 13) { K = 13; printf("K should not be less than 13\n"); }
	else if (K > max_k) K = max_k;

	time_t t_start = time(NULL);
	g_t_last = g_t_main = now_ms();
	if (sdt_test_env("SDT_LAYOUT_CHECK")) setenv("SDT_KEEP_FIRST", "1", 0);      /* the check sorts by the first-occurrence ordinals once more */
	if (sdt_env("SDT_TIMING")) {
		/* how long the loader took to get here (process start from /proc/self/stat, in clock ticks since boot): what a caller's
		 * wall clock holds beyond "total inside main" is this plus the kernel's teardown of the address space after _exit */
		FILE *sf = fopen("/proc/self/stat", "r");
		char sb[2048];
		if (sf) {
			const size_t got = fread(sb, 1, sizeof sb - 1, sf);
			fclose(sf);
			sb[got] = 0;
			const char *q = strrchr(sb, ')');
			unsigned long long start = 0;
			int field = 2;
			for (q = q ? q + 1 : sb; *q && field < 22; q++) if (*q == ' ') field++;
			if (field == 22) start = strtoull(q, NULL, 10);
			struct timespec bt;
			clock_gettime(CLOCK_BOOTTIME, &bt);
			const double since = bt.tv_sec * 1e3 + bt.tv_nsec * 1e-6 - (double)start * 1e3 / (double)sysconf(_SC_CLK_TCK);
			fprintf(stderr, "[sdt-pregraph] %-28s %9.1f ms\n", "before main (loader)", since);
		}
	}
	sdt_cfg cfg;
	if (sdt_cfg_load(cfgfile, &cfg) != 0) return 255;
	int max_read_len = cfg.max_rd_len ? cfg.max_rd_len : 100;                /* prlHashReads.c:361-364 */
	printf("In %s, %d libs, max seq len %d, max name len %d\n\n", cfgfile, cfg.nlibs, max_read_len, 256);

	/* --gpus N: one process per GPU, forked BEFORE anything touches the HIP runtime.  Rank 0 is this process: it prints,
	 * writes the files and runs the graph phases; the others count their share of the reads and hand their nodes over. */
	boot_t *boot = NULL;
	if (gpus < 1 || gpus > 64) { fprintf(stderr, "--gpus must be 1..64\n"); return 255; }
	if (gpus > 1) {
		boot = (boot_t *)mmap(NULL, sizeof(boot_t), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
		if (boot == MAP_FAILED) { perror("mmap"); return 1; }
		memset(boot, 0, sizeof *boot);
		snprintf(boot->name, sizeof boot->name, "pg%d", (int)getpid());
		snprintf(g_shm_name, sizeof g_shm_name, "%s", boot->name);
		g_shm_ranks = gpus;
		fflush(stdout);
		struct sigaction sa;
		memset(&sa, 0, sizeof sa);
		sa.sa_handler = on_sigchld;
		sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
		sigaction(SIGCHLD, &sa, NULL);
		atexit(kill_children);
		const pid_t parent = getpid();
		for (int r = 1; r < gpus; r++) {
			/* SIGCHLD stays blocked from before the fork until the pid is on record: a child that dies at once is then reaped
			 * by the handler like any other (not left as a slot that is waited for at the end and killed by a recycled pid) */
			sigset_t blk, old;
			sigemptyset(&blk);
			sigaddset(&blk, SIGCHLD);
			sigprocmask(SIG_BLOCK, &blk, &old);
			const pid_t pid = fork();
			if (pid < 0) { perror("fork"); return 1; }
			if (pid == 0) {
				rank = r;
				g_quiet = 1;
				g_nchild = 0;                                         /* (a child has no children to take along) */
				g_shm_name[0] = 0;                                    /* (nor the job's segments to clear away) */
				signal(SIGCHLD, SIG_DFL);
				sigprocmask(SIG_SETMASK, &old, NULL);
				prctl(PR_SET_PDEATHSIG, SIGKILL);
				if (getppid() != parent) return 1;                    /* the parent is gone already (also under a subreaper) */
				if (!freopen("/dev/null", "w", stdout)) return 1;       /* one voice: rank 0's */
				break;
			}
			g_child[g_nchild] = pid;
			g_nchild = g_nchild + 1;
			sigprocmask(SIG_SETMASK, &old, NULL);
		}
	}
	/* parser threads: the job's CPUs minus the pushing thread and the runtime's helpers -- under a CPU quota (cgroup cpu.max) one
	 * runnable thread too many throttles every thread of the process, the one that feeds the device included (200 M reads with
	 * -p 16 on 16 CPUs: 6.3 s against 1.6 s with -p 8).  -p stays the number of sets of the layout (graph.c). */
	int parse_threads = threads;
	{
		const int usable = par_threads();
		const int cap = sdt_env("SDT_PARSE_THREADS") ? atoi(sdt_env("SDT_PARSE_THREADS")) : (usable > 4 ? usable - 3 : usable);
		if (parse_threads > cap) parse_threads = cap > 0 ? cap : 1;
	}
	/* --gpus N: every rank parses its own chunks and nothing else (seqio.h: sdt_read_shard_skip_foreign), so the parser threads are
	 * shared out evenly -- two at least; when one rank keeps every read (SDT_RANK0_MAP, the way of rounds 2-4) rank 0 parses all of the
	 * text and gets most of the threads, the others only count records */
	const int rank0_keeps_all = gpus > 1 && !hash_only && !host_map && !(!host_walks && threads <= 256 && !sdt_test_env("SDT_HOST_LAYOUT") && !sdt_test_env("SDT_RANK0_MAP"));
	const int my_threads = gpus == 1 ? parse_threads
	                     : rank0_keeps_all ? (rank == 0 ? (threads - (gpus - 1) > threads / 2 ? threads - (gpus - 1) : (threads + 1) / 2) : 2)
	                     : (parse_threads / gpus > 2 ? parse_threads / gpus : 2);
	sdt_ctx *gpu = NULL;
	/* SDT_PIPELINE=1 (tests): the locality pipeline also for jobs below its 2^27 k-mer threshold */
	const uint32_t iflags = (hash_only ? 0 : (SDT_FLAG_TRACK_FIRST | ((host_map || gpus > 1) ? 0 : SDT_FLAG_KEEP_READS))) |
	                        (sdt_test_env("SDT_PIPELINE") ? SDT_FLAG_PARTITION : 0);
	if (sdt_gpu_init(&gpu, share_device ? device : device + rank, K, est / (unsigned long long)gpus, iflags) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_init: %s\n", sdt_gpu_last_error());
		return 1;
	}
	if (gpus > 1) {
		int rcc;
		if (share_device) {
			rcc = sdt_gpu_comm_init_shm(gpu, boot->name, rank, gpus);
		} else {
			if (rank == 0) {
				rcc = sdt_gpu_comm_id(&boot->id);
				__sync_synchronize();
				boot->ready = rcc == SDT_OK ? 1 : -1;
			}
			while (!boot->ready) usleep(1000);
			rcc = boot->ready == 1 ? sdt_gpu_comm_init(gpu, &boot->id, rank, gpus) : SDT_EHIP;
		}
		if (rcc != SDT_OK) { fprintf(stderr, "[rank %d] communicator: %s\n", rank, sdt_gpu_last_error()); return 1; }
	}
	phase("config + gpu init");
	push_state st;
	memset(&st, 0, sizeof st);
	/* --gpus N: every rank keeps the reads it parsed and maps them itself once rank 0 has the graph (the default path: layout, cutting
	 * and edges on rank 0's device); with --host-map / --host-walks the second pass is the host's, which reads the files again */
	const int per_rank_map = gpus > 1 && !hash_only && !host_map && !host_walks && threads <= 256 && !sdt_test_env("SDT_HOST_LAYOUT") && !sdt_test_env("SDT_RANK0_MAP");
	st.gpu = gpu; st.rank = rank; st.nranks = gpus; st.keep_all = gpus > 1 && rank == 0 && !hash_only && !host_map && !per_rank_map;
	st.keep_mine = per_rank_map;
	const size_t chunk = sdt_test_env("SDT_CHUNK_BYTES") ? (size_t)strtoull(sdt_test_env("SDT_CHUNK_BYTES"), NULL, 10) : (size_t)(32u << 20);   /* (tests: many small chunks) */
	int rc;
	uint64_t text_parsed = 0, text_seen = 0;
	if (gpus == 1) {
		st.K = K;
		st.total_text = input_bytes(&cfg);
		if (!sdt_tuning_env("SDT_NO_PINNED_POOL")) sdt_pool_enable(sdt_gpu_host_alloc, sdt_gpu_host_free, my_threads + PUSH_DEPTH + 8);
		rc = sdt_stream_reads(&cfg, max_read_len, my_threads, chunk, 1, push_batch, &st, NULL);
		if (rc == 0 && inflight_retire(gpu, 0) != 0) rc = -1;
		sdt_pool_disable();
	} else {
		sdt_read_shard_begin(rank, gpus, st.keep_all);
		/* (the same on every rank: only rank 0 has keep_all set, but whether ANY rank keeps everything is a property of the run) */
		st.defer = hash_only || host_map || per_rank_map;
		sdt_read_shard_skip_foreign(st.defer);
		sdt_stream_ordinals_init(&st.ords);
		rc = sdt_stream_reads(&cfg, max_read_len, my_threads, chunk, 1, push_batch_sharded, &st, NULL);
		if (rc == 0 && st.fill) rc = flush_group(&st);
		text_parsed = sdt_reader_bytes_parsed; text_seen = sdt_reader_bytes_seen;
		sdt_read_shard_begin(0, 1, 0);
	}
	if (rc != 0) { sdt_gpu_destroy(gpu); return 1; }
	uint64_t kmers = 0, nodes = 0, removed = 0, linear = 0;
	if (sdt_gpu_finish_count(gpu, &kmers, &nodes) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_finish_count: %s\n", sdt_gpu_last_error());
		return 1;
	}
	const uint64_t my_nodes = nodes;
	if (gpus > 1) {                                              /* the counters the reference prints are sums over its sets */
		int64_t v[2] = {(int64_t)kmers, (int64_t)nodes};
		if (sdt_gpu_allreduce_i64(gpu, v, 2) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
		kmers = (uint64_t)v[0]; nodes = (uint64_t)v[1];
	}
	if (sdt_env("SDT_TIMING") && !g_quiet) {
		double ms[SDT_NSTAGES];
		uint64_t cn[SDT_NCOUNTERS];
		if (sdt_gpu_stage_times(gpu, ms, cn) == SDT_OK)
			fprintf(stderr, "[ingest] consumer waited %.0f ms for the parsers, spent %.0f ms pushing; device stages: direct %.0f, scatter %.0f, split %.0f, count %.0f ms; %llu batches counted, %llu early flushes, %d parser threads\n",
			        sdt_reader_wait_ms, sdt_reader_fn_ms, ms[0], ms[1], ms[2], ms[3], (unsigned long long)cn[6], (unsigned long long)cn[3], my_threads);
	}
	if (sdt_env("SDT_TIMING") && !g_quiet && gpus > 1)
		fprintf(stderr, "[ingest] rank 0 of %d parsed %.1f of %.1f MB of text (%.3f of the input; the other chunks are their owners')\n", gpus,
		        text_parsed / 1e6, text_seen / 1e6, text_seen ? (double)text_parsed / (double)text_seen : 0.0);
	phase("parse + hash (GPU)");
	printf("time spent on hash reads: %ds, %llu reads processed\n", (int)(time(NULL) - t_start), st.reads);
	printf("%llu nodes allocated, %llu kmer in reads, %llu kmer processed\n", (unsigned long long)nodes,
	       (unsigned long long)kmers, (unsigned long long)kmers);
	if (d) {
		if (sdt_gpu_delow(gpu, d, &removed) != SDT_OK) { fprintf(stderr, "sdt_gpu_delow: %s\n", sdt_gpu_last_error()); return 1; }
	}
	int64_t hist[257];
	if (sdt_gpu_mark_and_hist(gpu, hist, &linear) != SDT_OK) {
		fprintf(stderr, "sdt_gpu_mark_and_hist: %s\n", sdt_gpu_last_error());
		return 1;
	}
	if (gpus > 1) {
		int64_t v[259];
		memcpy(v, hist, sizeof hist);
		v[257] = (int64_t)linear; v[258] = (int64_t)removed;
		if (sdt_gpu_allreduce_i64(gpu, v, 259) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
		memcpy(hist, v, sizeof hist);
		linear = (uint64_t)v[257]; removed = (uint64_t)v[258];
	}
	if (d) printf("%llu kmer removed\n", (unsigned long long)removed);
	printf("%llu linear nodes\n", (unsigned long long)linear);
	char name[4200];
	snprintf(name, sizeof name, "%s.kmerFreq", prefix);
	if (rank == 0) {
		FILE *fo = fopen(name, "w");
		if (!fo) { printf("Cannot open %s. Now exit to system...\n", name); return 255; }
		for (int i = 1; i < 256; i++)
			fprintf(fo, "%lld\n", (long long)hist[i]);
		fclose(fo);
	}
	phase("delow/mark/kmerFreq (GPU)");
	printf("time spent on pre-graph construction: %ds\n\n", (int)(time(NULL) - t_start));
	printf("deLowKmer %d, deLowEdge %d\n", d, 1);
	if (!hash_only) {
		/* hand the node table to the host graph phases in the reference's visiting order (graph/graph.h) */
		uint64_t n = 0;
		const int nwk = sdt_gpu_key_words(gpu), nwv = max_k <= 31 ? 1 : (max_k <= 63 ? 2 : 4);
		if (sdt_gpu_export_nodes(gpu, NULL, NULL, NULL, NULL, NULL, 0, &n) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
		/* Past 2^32 - 16 nodes (the reference's sets are 64-bit: inc/newhash.h:79-88 `ubyte8 size, count, max`) the device's graph phases
		 * are out -- their walk records, component labels, layout ranks and edge records carry 32-bit node indices -- and the documented
		 * fallback takes over: the nodes go to the host in one export, the host replays the layout, builds its 64-bit index
		 * (graph_t.index64) and runs cutting and kmer2edges on its threads; pass 1 and the second read pass stay on the device (they
		 * address the table by key).  SDT_NODE_LIMIT moves the threshold so that the tests can take this path on a golden case. */
		const uint64_t node_limit = sdt_test_env("SDT_NODE_LIMIT") ? strtoull(sdt_test_env("SDT_NODE_LIMIT"), NULL, 10) : 0xFFFFFFF0ULL;
		if ((gpus > 1 ? nodes : n) >= node_limit) {
			if (!g_quiet) fprintf(stderr, "[sdt-pregraph] %llu nodes: past the 32-bit node indices of the device's graph phases; layout, cutting and edges run on the host\n",
			                      (unsigned long long)(gpus > 1 ? nodes : n));
			host_walks = 1;
			graph_force_wide_index = 1;
		}
		uint64_t *keys, *first = NULL;
		uint32_t *ll, *rf, *cnt;
		graph_t *G = NULL;
		dev_state *Dp = (dev_state *)calloc(1, sizeof(dev_state));
		Dp->nwk = nwk;
		int keys_in_device = 0;                                /* --gpus N: every shard is in rank 0's device table already */
		keys = NULL; ll = rf = cnt = NULL;
		if (gpus > 1) {
			/* shards -> rank 0.  Every rank learns all shard sizes; ranks > 0 export into a shared-memory segment each and
			 * leave once rank 0 has taken their nodes (host arrays + its own device table: sdt_gpu_import_nodes) */
			int64_t sizes[64];
			memset(sizes, 0, sizeof sizes);
			sizes[rank] = (int64_t)my_nodes;
			if (sdt_gpu_allreduce_i64(gpu, sizes, gpus) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
			const size_t per_node = (size_t)nwk * 8 + 8 + 12;
			char seg[128];
			int64_t token = 0;
			if (rank > 0) {
				snprintf(seg, sizeof seg, "/sdt_%s_n%d", boot->name, rank);
				uint8_t *m = (uint8_t *)shm_region(seg, (size_t)(my_nodes + 1) * per_node, 1);
				if (!m) { fprintf(stderr, "[rank %d] shared memory for %llu nodes failed\n", rank, (unsigned long long)my_nodes); return 1; }
				uint64_t *k2 = (uint64_t *)m, *f2 = k2 + (my_nodes + 1) * (size_t)nwk;
				uint32_t *l2 = (uint32_t *)(f2 + my_nodes + 1), *r2 = l2 + my_nodes + 1, *c2 = r2 + my_nodes + 1;
				if (sdt_gpu_export_nodes(gpu, k2, l2, r2, c2, f2, my_nodes, &n) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
				if (sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK ||          /* "my shard is in shared memory" */
				    sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK) {          /* "rank 0 has it" */
					fprintf(stderr, "[rank %d] %s\n", rank, sdt_gpu_last_error());
					return 1;
				}
				munmap(m, (size_t)(my_nodes + 1) * per_node);
				shm_unlink(seg);
				/* the shard is with rank 0: its table, the ordinals and the pools of the locality pipeline go back to the device now, not
				 * when the graph arrives (under --share-device they would sit beside rank 0's graph buffers until then) */
				if (sdt_gpu_release_table(gpu) != SDT_OK) { fprintf(stderr, "[rank %d] %s\n", rank, sdt_gpu_last_error()); return 1; }
				if (per_rank_map) {
					/* wait for the graph (rank 0 lays it out, cuts it and builds the edges: seconds to minutes -- polled, not a collective with
					 * its deadline; the parent's death takes this process along), map my reads, leave my arcs */
					while (boot->paths_state == 0) usleep(2000);
					int ok = boot->paths_state == 1;
					if (ok) {
						const uint64_t pn = boot->paths_n, qn = boot->patch_n;
						snprintf(seg, sizeof seg, "/sdt_%s_paths", boot->name);
						const size_t pbytes = (size_t)(pn + 1) * ((size_t)nwk * 8 + 8) + (size_t)(qn + 1) * ((size_t)nwk * 8 + 8);
						uint8_t *pm = (uint8_t *)shm_region(seg, pbytes, 0);
						uint64_t nreads2 = 0, narcs = 0;
						/* a rank that owned no chunk of the input (fewer chunks than ranks) kept no reads: it has nothing to map and leaves an
						 * empty arc list (sdt_gpu_map_reads would refuse: "the reads were not kept") */
						const int have_reads = st.kept_reads > 0;
						if (!pm) { fprintf(stderr, "[rank %d] cannot map the path table\n", rank); ok = 0; }
						if (ok && !have_reads) munmap(pm, pbytes);
						if (ok && have_reads) {
							const uint64_t *pk = (const uint64_t *)pm, *pw = pk + (pn + 1) * (size_t)nwk, *qk = pw + pn + 1, *qi = qk + (qn + 1) * (size_t)nwk;
							if (sdt_gpu_import_paths(gpu, pk, pw, pn, qk, qi, qn, boot->num_ed) != SDT_OK ||
							    sdt_gpu_map_reads(gpu, &nreads2, &narcs) != SDT_OK) {
								fprintf(stderr, "[rank %d] second pass: %s\n", rank, sdt_gpu_last_error());
								ok = 0;
							}
							munmap(pm, pbytes);
						}
						uint8_t *am = NULL;
						size_t abytes = 0;
						if (ok) {
							snprintf(seg, sizeof seg, "/sdt_%s_a%d", boot->name, rank);
							abytes = (size_t)(narcs + 1) * 20;
							am = (uint8_t *)shm_region(seg, abytes, 1);
							uint64_t *ao = (uint64_t *)am;
							uint32_t *af = am ? (uint32_t *)(ao + narcs + 1) : NULL, *at2 = af ? af + narcs + 1 : NULL, *amu = at2 ? at2 + narcs + 1 : NULL;
							if (!am || (have_reads && sdt_gpu_export_arcs(gpu, af, at2, amu, ao, narcs, &narcs) != SDT_OK)) {
								fprintf(stderr, "[rank %d] arcs: %s\n", rank, am ? sdt_gpu_last_error() : "no shared memory");
								ok = 0;
							}
						}
						boot->arcs_n[rank] = narcs;
						boot->arcs_reads[rank] = nreads2;
						__sync_synchronize();
						boot->arcs_state[rank] = ok ? 1 : -1;
						while (ok && boot->paths_state == 1) usleep(2000);     /* rank 0 is reading them */
						if (am) { munmap(am, abytes); shm_unlink(seg); }
					}
					sdt_gpu_destroy(gpu);
					return ok || boot->paths_state == -1 ? 0 : 1;
				}
				sdt_gpu_destroy(gpu);
				return 0;
			}
			n = nodes;                                            /* all shards */
			const int device_layout = !host_map && !host_walks && threads <= 256 && n < node_limit && !sdt_test_env("SDT_HOST_LAYOUT");
			keys_in_device = device_layout;
			if (device_layout) {
				/* the shards go straight into rank 0's device table (it then lays the whole graph out like a single-GPU run: below);
				 * nothing is gathered in host arrays */
				if (sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
				for (int r = 1; r < gpus; r++) {
					const uint64_t m_n = (uint64_t)sizes[r];
					snprintf(seg, sizeof seg, "/sdt_%s_n%d", boot->name, r);
					uint8_t *m = (uint8_t *)shm_region(seg, (size_t)(m_n + 1) * per_node, 0);
					if (!m) { fprintf(stderr, "cannot map the shard of rank %d\n", r); return 1; }
					const uint64_t *k2 = (const uint64_t *)m, *f2 = k2 + (m_n + 1) * (size_t)nwk;
					const uint32_t *l2 = (const uint32_t *)(f2 + m_n + 1), *r2 = l2 + m_n + 1, *c2 = r2 + m_n + 1;
					if (sdt_gpu_import_nodes(gpu, k2, l2, r2, c2, f2, m_n) != SDT_OK) { fprintf(stderr, "sdt_gpu_import_nodes: %s\n", sdt_gpu_last_error()); return 1; }
					munmap(m, (size_t)(m_n + 1) * per_node);
				}
				if (sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }   /* the other ranks may go */
				phase("shards into rank 0's table");
			} else {
			keys = (uint64_t *)malloc((n + 1) * (size_t)nwk * 8); first = (uint64_t *)malloc((n + 1) * 8);
			ll = (uint32_t *)malloc((n + 1) * 4); rf = (uint32_t *)malloc((n + 1) * 4); cnt = (uint32_t *)malloc((n + 1) * 4);
			if (!keys || !first || !ll || !rf || !cnt) { fprintf(stderr, "out of host memory for %llu nodes\n", (unsigned long long)n); return 1; }
			uint64_t got = 0;
			if (sdt_gpu_export_nodes(gpu, keys, ll, rf, cnt, first, my_nodes, &got) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
			if (sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
			uint64_t at = my_nodes;
			for (int r = 1; r < gpus; r++) {
				const uint64_t m_n = (uint64_t)sizes[r];
				snprintf(seg, sizeof seg, "/sdt_%s_n%d", boot->name, r);
				uint8_t *m = (uint8_t *)shm_region(seg, (size_t)(m_n + 1) * per_node, 0);
				if (!m) { fprintf(stderr, "cannot map the shard of rank %d\n", r); return 1; }
				const uint64_t *k2 = (const uint64_t *)m, *f2 = k2 + (m_n + 1) * (size_t)nwk;
				const uint32_t *l2 = (const uint32_t *)(f2 + m_n + 1), *r2 = l2 + m_n + 1, *c2 = r2 + m_n + 1;
				memcpy(keys + at * nwk, k2, m_n * (size_t)nwk * 8); memcpy(first + at, f2, m_n * 8);
				memcpy(ll + at, l2, m_n * 4); memcpy(rf + at, r2, m_n * 4); memcpy(cnt + at, c2, m_n * 4);
				at += m_n;
				munmap(m, (size_t)(m_n + 1) * per_node);
			}
			if (sdt_gpu_allreduce_i64(gpu, &token, 1) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }   /* the other ranks may go */
			if (at != n) { fprintf(stderr, "shards hold %llu nodes, the counters say %llu\n", (unsigned long long)at, (unsigned long long)n); return 1; }
			if (!host_map && sdt_gpu_import_nodes(gpu, keys + my_nodes * nwk, ll + my_nodes, rf + my_nodes, cnt + my_nodes, first + my_nodes, n - my_nodes) != SDT_OK) {
				fprintf(stderr, "sdt_gpu_import_nodes: %s\n", sdt_gpu_last_error());
				return 1;
			}
			}
		}
		if ((gpus == 1 || keys_in_device) && !host_map && !host_walks && threads <= 256 && n < node_limit && !sdt_test_env("SDT_HOST_LAYOUT")) {
			/* the visiting order with the device: it sorts the nodes by (set, first occurrence) and sends the keys, the host
			 * replays the probing of every set (graph_replay_order), the device numbers the nodes and sends them in that order */
			keys = (uint64_t *)malloc((n + 1) * (size_t)nwk * 8);
			uint64_t *set_start = (uint64_t *)calloc((size_t)threads + 1, sizeof(uint64_t));
			/* all of it on the device (sort, replay of the probing as rounds of priority insertion, numbering); the two-step form
			 * with the host's replay when a limit of the device form is passed, on request (SDT_HOST_REPLAY), and -- SDT_LAYOUT_CHECK
			 * -- beside it: both orders must then name the same key at every visiting position */
			int on_device = 0;
			uint64_t *check_keys = NULL;
			if (!sdt_test_env("SDT_HOST_REPLAY")) {
				const int rcl = sdt_gpu_layout_on_device(gpu, threads, nwv, graph_init_kmerset_size != 0, set_start, &n);
				if (rcl == SDT_OK) on_device = 1;
				else if (rcl == SDT_ELIMIT) { if (!g_quiet) fprintf(stderr, "[sdt-pregraph] %s: the host replays the layout\n", sdt_gpu_last_error()); }
				else if (rcl != SDT_EINVAL) { fprintf(stderr, "sdt_gpu_layout_on_device: %s\n", sdt_gpu_last_error()); return 1; }
				if (on_device) phase("layout: sort + replay + numbering (GPU)");
			}
			if (!on_device || sdt_test_env("SDT_LAYOUT_CHECK")) {
				uint64_t *hk = on_device ? (uint64_t *)malloc((n + 1) * (size_t)nwk * 8) : keys;
				uint64_t *ss = on_device ? (uint64_t *)calloc((size_t)threads + 1, sizeof(uint64_t)) : set_start;
				if (sdt_gpu_layout_sorted_keys(gpu, threads, nwv, hk, n, ss, &n) != SDT_OK) { fprintf(stderr, "sdt_gpu_layout_sorted_keys: %s\n", sdt_gpu_last_error()); return 1; }
				phase("layout: sort (GPU) + keys D2H");
				uint64_t *order = (uint64_t *)malloc((n + 1) * 8);
				graph_replay_order(nwv, nwk, threads, hk, ss, order);
				phase("layout: replay (host)");
				if (on_device) {
					check_keys = (uint64_t *)malloc((n + 1) * (size_t)nwk * 8);
					for (uint64_t vv = 0; vv < n; vv++) memcpy(check_keys + vv * nwk, hk + order[vv] * nwk, (size_t)nwk * 8);
					free(hk); free(ss);
				} else if (sdt_gpu_layout_apply(gpu, order, n) != SDT_OK) { fprintf(stderr, "sdt_gpu_layout_apply: %s\n", sdt_gpu_last_error()); return 1; }
				free(order);
			}
			ll = (uint32_t *)malloc((n + 1) * 4); rf = (uint32_t *)malloc((n + 1) * 4); cnt = (uint32_t *)malloc((n + 1) * 4);
			if (sdt_gpu_export_ordered(gpu, keys, ll, rf, cnt, n) != SDT_OK) { fprintf(stderr, "sdt_gpu_export_ordered: %s\n", sdt_gpu_last_error()); return 1; }
			phase("layout: export in visiting order (D2H)");
			if (check_keys) {
				if (memcmp(check_keys, keys, n * (size_t)nwk * 8) != 0) { fprintf(stderr, "SDT_LAYOUT_CHECK: the device's visiting order differs from the host replay's\n"); return 1; }
				fprintf(stderr, "[sdt-pregraph] SDT_LAYOUT_CHECK: device and host replay agree on all %llu visiting positions\n", (unsigned long long)n);
				free(check_keys);
			}
			Dp->gpu = gpu; Dp->indexed = 1; Dp->by_index = 1;
			graph_index_hook = dev_index_hook;
			graph_index_hook_user = Dp;
			graph_index_hook_early = sdt_tuning_env("SDT_INDEX_INLINE") == NULL;      /* (the device numbered the nodes itself) */
			G = graph_from_ordered(K, nwv, nwk, threads, n, keys, ll, rf, cnt, set_start);
			graph_free_later(keys, ll, rf, cnt);
			free(set_start);
			phase("graph + index");
		} else if (gpus == 1) {
			keys = (uint64_t *)malloc((n + 1) * (size_t)nwk * 8); first = (uint64_t *)malloc((n + 1) * 8);
			ll = (uint32_t *)malloc((n + 1) * 4); rf = (uint32_t *)malloc((n + 1) * 4); cnt = (uint32_t *)malloc((n + 1) * 4);
			if (sdt_gpu_export_nodes(gpu, keys, ll, rf, cnt, first, n, &n) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
		}
		if (!G) {
			if (host_map) { sdt_gpu_destroy(gpu); gpu = NULL; }
			phase("export nodes (D2H)");
			Dp->gpu = gpu;
			if (gpu && !host_walks) {
				graph_index_hook = dev_index_hook;
				graph_index_hook_user = Dp;
			}
			G = graph_build(K, nwv, nwk, threads, n, keys, ll, rf, cnt, first);
			graph_free_later(keys, first, ll, rf);
			free(cnt);
			phase("layout replay + index (host)");
		}
		if (gpu && !host_walks) {                                          /* dry runs from the device mirror of the graph */
			G->dirty = (uint8_t *)calloc(G->n + 1, 1);
			G->dev_walks = dev_walks_hook;
			G->dev_minor_out = dev_minor_out_hook;
			if (Dp->by_index) { G->dev_minor_out_commit_begin = dev_minor_out_commit_begin_hook; G->dev_minor_out_commit_finish = dev_minor_out_commit_finish_hook; }
			G->dev_edge_ports = dev_edge_ports_hook;
			if (Dp->by_index && !sdt_test_env("SDT_HOST_EDGES")) G->dev_build_edges = dev_build_edges_hook;
			G->dev_user = Dp;
		}
		uint64_t nv_early = 0;
		int have_nv = 0;
		pthread_t vth_keep;
		vx_job *vj_keep = NULL;
		memset(&vth_keep, 0, sizeof vth_keep);
		time_t t0 = time(NULL);
		graph_remove_minor_out(G, dd);                                     /* pregraph.c:68-71 */
		phase(G->dev_minor_out ? "removeMinorOut (GPU dry run + host commit)" : "removeMinorOut (host)");
		printf("time spent on cut kmer: %ds\n\n", (int)(time(NULL) - t0));
		t0 = time(NULL);
		if (!d) graph_remove_single_tips(G);                               /* pregraph.c:75-88 */
		graph_remove_minor_tips(G);
		phase(G->dev_walks ? "tip cutting (GPU walks + host commit)" : "tip cutting (host)");
		printf("time spent on cutTipe: %ds\n\n", (int)(time(NULL) - t0));
		t0 = time(NULL);
		uint64_t ne = graph_build_edges(G, prefix);                        /* pregraph.c:95-98 */
		phase(G->dev_edge_ports ? "kmer2edges (GPU walks + host ids, stamping)" : "kmer2edges (host)");
		printf("time spent on making edges: %ds\n\n", (int)(time(NULL) - t0));
		t0 = time(NULL);
		printf("%d thread created prlRead2path\n", threads);                /* pregraph.c:101-104 */
		const int rank0_host_map = per_rank_map && !keys_in_device;       /* (the graph did not take the device path: rank 0 holds only its own reads) */
		if (gpus > 1 && per_rank_map && rank0_host_map) { __sync_synchronize(); boot->paths_state = -1; }
		if (host_map || rank0_host_map) {
			arc_state as = {G, arcs_new(), 0};
			if (sdt_stream_reads(&cfg, max_read_len, threads, chunk, 1, arc_batch, &as, NULL) != 0) return 1;
			printf("%llu reads processed\n", as.reads);
			arcs_write(as.A, prefix);
			arcs_free(as.A);
		} else {
			/* second pass on the GPU over the reads kept in HBM: send the cleaned graph back as path words */
			/* with the device mirror in place the path words go over by node index; otherwise with their keys */
			const int by_index = G->dev_walks != NULL && Dp->indexed;
			/* the edges were built on the device: the path words are there already */
			uint64_t *pk = NULL, *pw = NULL;
			if (!G->edges_on_device) {
				pk = by_index ? NULL : (uint64_t *)malloc((G->n + 1) * (size_t)nwk * 8);
				pw = (uint64_t *)malloc((G->n + 1) * 8);
				void *pa[4] = {G, pk, pw, (void *)(intptr_t)nwk};
				par_for(0, G->n, 1 << 16, gather_paths, pa);
			}
			uint64_t np = 0;
			uint64_t *qk = (uint64_t *)malloc((G->patch_n + 1) * (size_t)nwk * 8), *qi = (uint64_t *)malloc((G->patch_n + 1) * 8);
			for (uint64_t i = 0; G->patch && i <= G->patch_mask; i++)
				if (G->patch[i].used) {
					for (int w = 0; w < nwk; w++) qk[np * nwk + w] = G->patch[i].seq.w[4 - nwk + w];
					qi[np++] = (uint64_t)G->patch[i].edge | ((uint64_t)G->patch[i].twin << 32);
				}
			/* with the edges (and the path words) made on the device the host graph has one duty left, *.vertex: write it now and
			 * let go of the node array while the device maps the reads (its line is printed in its turn) */
			pthread_t vth;
			vx_job VJ = {G, prefix, 0};
			if (G->edges_on_device) {
				graph_vertex_quiet = 1;
				vj_keep = (vx_job *)malloc(sizeof VJ);
				*vj_keep = VJ;
				have_nv = pthread_create(&vth, NULL, vertex_thread, vj_keep) == 0;
				if (!have_nv) graph_vertex_quiet = 0;                 /* (no thread: *.vertex is written in its turn below, with its line) */
				vth_keep = vth;
			}
			uint64_t nreads2 = 0, narcs = 0;
			const double t_r0 = now_ms();
			if (sdt_gpu_load_paths(gpu, pk, pw, G->n, qk, qi, np, G->num_ed) != SDT_OK) {
				fprintf(stderr, "second pass: %s\n", sdt_gpu_last_error());
				return 1;
			}
			uint8_t *paths_m = NULL;
			size_t paths_bytes = 0;
			char paths_seg[128] = "";
			if (gpus > 1 && per_rank_map) {
				/* every rank maps its own reads: the graph as that pass needs it goes into shared memory */
				uint64_t pn = 0;
				if (sdt_gpu_export_paths(gpu, NULL, NULL, 0, &pn) != SDT_OK) { fprintf(stderr, "second pass: %s\n", sdt_gpu_last_error()); return 1; }
				snprintf(paths_seg, sizeof paths_seg, "/sdt_%s_paths", boot->name);
				paths_bytes = (size_t)(pn + 1) * ((size_t)nwk * 8 + 8) + (size_t)(np + 1) * ((size_t)nwk * 8 + 8);
				paths_m = (uint8_t *)shm_region(paths_seg, paths_bytes, 1);
				if (!paths_m) { fprintf(stderr, "shared memory for the path table of %llu nodes failed\n", (unsigned long long)pn); return 1; }
				uint64_t *sk = (uint64_t *)paths_m, *sw = sk + (pn + 1) * (size_t)nwk, *sqk = sw + pn + 1, *sqi = sqk + (np + 1) * (size_t)nwk;
				if (sdt_gpu_export_paths(gpu, sk, sw, pn, &pn) != SDT_OK) { fprintf(stderr, "second pass: %s\n", sdt_gpu_last_error()); return 1; }
				memcpy(sqk, qk, np * (size_t)nwk * 8);
				memcpy(sqi, qi, np * 8);
				boot->paths_n = pn; boot->patch_n = np; boot->num_ed = G->num_ed;
				__sync_synchronize();
				boot->paths_state = 1;
				phase("path table -> shared memory");
			}
			const double t_r1 = now_ms();
			if (sdt_gpu_map_reads(gpu, &nreads2, &narcs) != SDT_OK) {
				fprintf(stderr, "second pass: %s\n", sdt_gpu_last_error());
				return 1;
			}
			const double t_r2 = now_ms();
			free(pk); free(pw); free(qk); free(qi);
			uint32_t *af = (uint32_t *)malloc((narcs + 1) * 4), *at = (uint32_t *)malloc((narcs + 1) * 4), *am = (uint32_t *)malloc((narcs + 1) * 4);
			uint64_t *ao = (uint64_t *)malloc((narcs + 1) * 8);
			if (sdt_gpu_export_arcs(gpu, af, at, am, ao, narcs, &narcs) != SDT_OK) { fprintf(stderr, "%s\n", sdt_gpu_last_error()); return 1; }
			if (gpus > 1 && per_rank_map) {
				/* the arcs of the other ranks: same (from, to) pairs add up, the first occurrence is the earliest (prlRead2path.c:415-430
				 * counts per thread and adds up the same way) */
				uint64_t total = narcs;
				for (int r = 1; r < gpus; r++) {
					while (boot->arcs_state[r] == 0) usleep(1000);
					if (boot->arcs_state[r] < 0) { fprintf(stderr, "rank %d failed in the second pass\n", r); return 1; }
					total += boot->arcs_n[r];
					nreads2 += boot->arcs_reads[r];
				}
				af = (uint32_t *)realloc(af, (total + 1) * 4); at = (uint32_t *)realloc(at, (total + 1) * 4); am = (uint32_t *)realloc(am, (total + 1) * 4);
				ao = (uint64_t *)realloc(ao, (total + 1) * 8);
				if (!af || !at || !am || !ao) { fprintf(stderr, "out of host memory for %llu arcs\n", (unsigned long long)total); return 1; }
				uint64_t at_ = narcs;
				for (int r = 1; r < gpus; r++) {
					const uint64_t m_n = boot->arcs_n[r];
					char seg2[128];
					snprintf(seg2, sizeof seg2, "/sdt_%s_a%d", boot->name, r);
					const size_t ab = (size_t)(m_n + 1) * 20;
					uint8_t *m2 = (uint8_t *)shm_region(seg2, ab, 0);
					if (!m2) { fprintf(stderr, "cannot map the arcs of rank %d\n", r); return 1; }
					const uint64_t *o2 = (const uint64_t *)m2;
					const uint32_t *f2 = (const uint32_t *)(o2 + m_n + 1), *t2 = f2 + m_n + 1, *u2 = t2 + m_n + 1;
					memcpy(ao + at_, o2, m_n * 8); memcpy(af + at_, f2, m_n * 4); memcpy(at + at_, t2, m_n * 4); memcpy(am + at_, u2, m_n * 4);
					at_ += m_n;
					munmap(m2, ab);
				}
				__sync_synchronize();
				boot->paths_state = 2;                                /* the other ranks may go */
				if (paths_m) { munmap(paths_m, paths_bytes); shm_unlink(paths_seg); }
				narcs = arcs_combine(af, at, am, ao, total);
			}
			const double t_r3 = now_ms();
			printf("%llu reads processed\n", (unsigned long long)nreads2);
			arcs_write_arrays(prefix, af, at, am, ao, narcs);
			if (sdt_env("SDT_TIMING") && !g_quiet)
				fprintf(stderr, "[read2edge] gather %.1f ms, load paths + patch table %.1f ms, map reads %.1f ms, export %llu arcs %.1f ms, sort + write %.1f ms\n",
				        t_r0 - g_t_last, t_r1 - t_r0, t_r2 - t_r1, (unsigned long long)narcs, t_r3 - t_r2, now_ms() - t_r3);
			free(af); free(at); free(am); free(ao);
		}
		phase(host_map ? "read2edge (host)" : "read2edge (GPU)");
		printf("time spent on mapping reads: %ds\n\n", (int)(time(NULL) - t0));
		graph_edges_join(G);                                                /* (*.edge.gz was written beside the second read pass) */
		if (have_nv) { pthread_join(vth_keep, NULL); nv_early = vj_keep->nv; }
		uint64_t nv = nv_early;
		if (have_nv) printf("%llu vertex outputed\n", (unsigned long long)nv);
		else nv = graph_write_vertex(G, prefix);                            /* pregraph.c:106 */
		graph_write_basic(prefix, nv, K, ne, max_read_len);
		phase("vertex + preGraphBasic");              /* G is not freed: the process ends here and the kernel is faster at it */
	}
	if (gpu) sdt_gpu_destroy(gpu);
	phase("release the device");
	if (sdt_env("SDT_TIMING") && !g_quiet) fprintf(stderr, "[sdt-pregraph] %-28s %9.1f ms\n", "total inside main", now_ms() - g_t_main);
	sdt_cfg_free(&cfg);
	if (gpus == 1 && !sdt_env("SDT_SLOW_EXIT")) {
		/* every file is closed and the device is released: skip the runtime's and the allocator's own teardown (atexit handlers,
		 * unloading code objects, returning gigabytes page by page) -- the kernel takes the address space back in one go */
		fflush(stdout);
		fflush(stderr);
		_exit(0);
	}
	if (gpus > 1 && rank == 0) {                                 /* the ranks that are still leaving (the handler reaps them) */
		for (int tries = 0; tries < 30000; tries++) {
			int alive = 0;
			for (int i = 0; i < g_nchild; i++) alive += g_child[i] > 0;
			if (!alive) break;
			usleep(1000);
		}
	}
	return 0;
}
