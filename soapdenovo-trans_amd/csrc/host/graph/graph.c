/* graph.c -- see graph.h: layout replay, flat node array, lookup index, hash_kmer. */
#include "graph.h"
#include <pthread.h>
#include <math.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include "par.h"
#include "big.h"        /* every large block (replay tables, node arrays) asks for transparent huge pages */

/* ---- hash_kmer: table-driven CRC-32 with a SIGNED 32-bit state (arithmetic >> 8), hashFunction.c:83-122 ---- */
static int32_t crc_tab[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;

static void crc_init(void)
{
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int b = 0; b < 8; b++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
		crc_tab[n] = (int32_t)c;
	}
}

uint64_t ref_hash_kmer(const kw_t *k, int nw)
{
	pthread_once(&crc_once, crc_init);               /* called from the unpack threads */
	/* raw bytes of the variant's Kmer struct: words high..low, each little endian */
	const unsigned char *raw = (const unsigned char *)&k->w[4 - nw];
	int32_t crc = ~0;
	for (int i = 0; i < 8 * nw; i++)
		crc = crc_tab[(crc ^ raw[i]) & 0xff] ^ (crc >> 8);       /* >> on a negative int: sign bits shift in */
	crc = ~crc;
	return (uint64_t)(int64_t)crc & 0xffffffULL;
}

/* ---- layout replay --------------------------------------------------------------------------------------
 * put_kmerset + encap_kmerset (newhash.c:293-462) reduced to what fixes the slot of every key: a table of
 * node ids.  size: "prime" >= request found by trial division with bound i < (ubyte8)sqrt((float)n)
 * (:116-158); max = size * 0.77f at init (float product, :176) and size * (double)0.77f after growth (:350);
 * growth doubles below 2^28-1 else adds 0xFFFFFF, until n * lf >= count + 1 (:318-330); the rehash is IN PLACE:
 * old entries are visited by slot, each is re-probed in the new geometry and an occupant that has not moved
 * yet is carried onward (:359-406). */
int graph_init_kmerset_size = 0;      /* -a (initKmerSetSize, pregraph.c:160-162); only the 2- and 4-word variants look at it */

static int prime_kh(uint64_t num)
{
	if (num < 4) return 1;
	if (num % 2 == 0) return 0;
	uint64_t lim = (uint64_t)sqrt((float)num);
	for (uint64_t i = 3; i < lim; i += 2)
		if (num % i == 0) return 0;
	return 1;
}
static uint64_t next_prime_kh(uint64_t n)
{
	if (n % 2 == 0) n++;
	while (!prime_kh(n)) n += 2;
	return n;
}

typedef struct {
	int64_t *slot;         /* node id or -1 */
	uint8_t *moved;        /* scratch for the rehash */
	uint64_t size, count, max;
	double lf;
} replay_t;

static uint64_t home_slot(const kw_t *k, int nw, uint64_t size)
{
	if (nw == 1) return k->w[3] % size;                                  /* newhash.c:428 */
	if (nw == 2) {                                                       /* :423-425 */
		unsigned __int128 v = ((unsigned __int128)k->w[2] << 64) | k->w[3];
		return (uint64_t)(v % size);
	}
	uint64_t t = k->w[0] % size;                                         /* :43-55, 32 bits at a time */
	const uint64_t part[6] = {k->w[1] >> 32, k->w[1] & 0xffffffffu, k->w[2] >> 32, k->w[2] & 0xffffffffu,
	                          k->w[3] >> 32, k->w[3] & 0xffffffffu};
	for (int i = 0; i < 6; i++)
		t = ((t << 32) | part[i]) % size;
	return t;
}

static void replay_grow(replay_t *r, const gnode_t *nodes, int nw)
{
	uint64_t n = r->size;
	do {
		n = n < 0xFFFFFFFu ? n << 1 : n + 0xFFFFFFu;
		n = next_prime_kh(n);
	} while (n * r->lf < (double)(r->count + 1));
	const uint64_t old = r->size;
	int64_t *ns = (int64_t *)malloc(n * sizeof(int64_t));
	/* the reference reallocs in place and tells old from new by two flag arrays; here: ns = new geometry,
	 * r->slot = old geometry, moved[i] = old entry i has left its old slot */
	for (uint64_t i = 0; i < n; i++) ns[i] = -1;
	r->moved = (uint8_t *)realloc(r->moved, old);
	memset(r->moved, 0, old);
	for (uint64_t i = 0; i < old; i++) {
		if (r->slot[i] < 0 || r->moved[i]) continue;
		int64_t carry = r->slot[i];
		r->moved[i] = 1;
		for (;;) {
			uint64_t h = home_slot(&nodes[carry].seq, nw, n);
			while (ns[h] >= 0) h = h + 1 == n ? 0 : h + 1;
			/* the new array aliases the old one below `old`: a slot h < old that still holds an unmoved old
			 * entry is "empty" in the new flags, and its occupant gets evicted and carried on */
			if (h < old && r->slot[h] >= 0 && !r->moved[h]) {
				const int64_t evicted = r->slot[h];
				r->moved[h] = 1;
				ns[h] = carry;
				carry = evicted;
				continue;
			}
			ns[h] = carry;
			break;
		}
	}
	free(r->slot);
	r->slot = ns;
	r->size = n;
	r->max = (uint64_t)(n * r->lf);
}

static void replay_put(replay_t *r, const gnode_t *nodes, int64_t id, int nw)
{
	if (r->count + 1 > r->max)
		replay_grow(r, nodes, nw);
	uint64_t h = home_slot(&nodes[id].seq, nw, r->size);
	while (r->slot[h] >= 0) h = h + 1 == r->size ? 0 : h + 1;
	r->slot[h] = id;
	r->count++;
}

/* ---- the same replay for 1-word keys with the key kept next to the node id: a rehash then walks the old slots
 * in order and touches one random cache line per entry (its new slot) instead of two (plus the 48-byte node) ---- */
typedef struct { int64_t id; uint64_t key; } rslot1;
typedef struct {
	rslot1 *slot;
	uint8_t *moved;
	uint64_t size, count, max;
	double lf;
} replay1_t;

static void replay1_grow(replay1_t *r)
{
	uint64_t n = r->size;
	do {
		n = n < 0xFFFFFFFu ? n << 1 : n + 0xFFFFFFu;
		n = next_prime_kh(n);
	} while (n * r->lf < (double)(r->count + 1));
	const uint64_t old = r->size;
	rslot1 *ns = (rslot1 *)malloc(n * sizeof(rslot1));
	for (uint64_t i = 0; i < n; i++) ns[i].id = -1;
	r->moved = (uint8_t *)realloc(r->moved, old);
	memset(r->moved, 0, old);
	for (uint64_t i = 0; i < old; i++) {
		/* the old slots are walked in order, so the new home of the entry a few steps ahead is known: have it in cache */
		if (i + 12 < old && r->slot[i + 12].id >= 0) {
			const uint64_t hp = r->slot[i + 12].key % n;
			__builtin_prefetch(&ns[hp], 1);
			if (hp < old) { __builtin_prefetch(&r->slot[hp]); __builtin_prefetch(&r->moved[hp], 1); }   /* the eviction test reads both */
		}
		if (r->slot[i].id < 0 || r->moved[i]) continue;
		rslot1 carry = r->slot[i];
		r->moved[i] = 1;
		for (;;) {
			uint64_t h = carry.key % n;
			while (ns[h].id >= 0) h = h + 1 == n ? 0 : h + 1;
			if (h < old && r->slot[h].id >= 0 && !r->moved[h]) {        /* see replay_grow */
				const rslot1 evicted = r->slot[h];
				r->moved[h] = 1;
				ns[h] = carry;
				carry = evicted;
				continue;
			}
			ns[h] = carry;
			break;
		}
	}
	free(r->slot);
	r->slot = ns;
	r->size = n;
	r->max = (uint64_t)(n * r->lf);
}

static inline void replay1_put(replay1_t *r, int64_t id, uint64_t key)
{
	if (r->count + 1 > r->max)
		replay1_grow(r);
	uint64_t h = key % r->size;
	while (r->slot[h].id >= 0) h = h + 1 == r->size ? 0 : h + 1;
	r->slot[h].id = id;
	r->slot[h].key = key;
	r->count++;
}

/* ---- our own lookup index ---- */
static inline uint64_t mix_key(const kw_t *k)
{
	uint64_t h = 0x9E3779B97F4A7C15ULL;
	for (int i = 0; i < 4; i++) {
		h ^= k->w[i];
		h ^= h >> 32; h *= 0xD6E8FEB86659FD93ULL; h ^= h >> 32;
	}
	return h;
}

gnode_t *graph_find_oriented(graph_t *g, kw_t word, int *smaller)
{
	kw_t bal = kw_rc(word, g->K);
	const kw_t *canon = &word;
	*smaller = 1;
	if (kw_less(&bal, &word)) {       /* KmerLarger(word, bal_word): keep the smaller strand */
		canon = &bal;
		*smaller = 0;
	}
	uint64_t h = mix_key(canon) & g->index_mask;
	for (;;) {
		const uint64_t v = g->index64 ? g->index64[h] : g->index[h];
		if (!v) break;
		if (kw_eq(&g->nodes[v - 1].seq, canon)) return &g->nodes[v - 1];
		h = (h + 1) & g->index_mask;
	}
	printf("kmer %llx %llx %llx %llx not found\n", (unsigned long long)canon->w[0], (unsigned long long)canon->w[1],
	       (unsigned long long)canon->w[2], (unsigned long long)canon->w[3]);
	exit(1);
}

#include <time.h>
static double gb_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
typedef struct { uint64_t first; uint64_t id; } ord_t;
static int cmp_ord(const void *a, const void *b)
{
	const ord_t *x = (const ord_t *)a, *y = (const ord_t *)b;
	return x->first < y->first ? -1 : x->first > y->first;
}

/* first-occurrence ordinals are unique, dense-ish 64-bit integers: LSD radix sort, 8 bits per pass, skipping the
 * bytes that are the same in every key (a qsort of the 23 M nodes of one set of a 50 M-read run took ~3 s) */
static void sort_ord(ord_t *a, uint64_t n)
{
	if (n < 4096) { qsort(a, (size_t)n, sizeof(ord_t), cmp_ord); return; }
	ord_t *tmp = (ord_t *)malloc((size_t)n * sizeof(ord_t));
	if (!tmp) { qsort(a, (size_t)n, sizeof(ord_t), cmp_ord); return; }
	uint64_t all_or = 0, all_and = ~(uint64_t)0;
	for (uint64_t i = 0; i < n; i++) { all_or |= a[i].first; all_and &= a[i].first; }
	const uint64_t varying = all_or ^ all_and;
	ord_t *src = a, *dst = tmp;
	for (int shift = 0; shift < 64; shift += 8) {
		if (!((varying >> shift) & 0xFF)) continue;
		uint64_t cnt[257] = {0};
		for (uint64_t i = 0; i < n; i++) cnt[((src[i].first >> shift) & 0xFF) + 1]++;
		for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
		for (uint64_t i = 0; i < n; i++) dst[cnt[(src[i].first >> shift) & 0xFF]++] = src[i];
		ord_t *t = src; src = dst; dst = t;
	}
	if (src != a) memcpy(a, src, (size_t)n * sizeof(ord_t));
	free(tmp);
}

/* ---- small fork/join helper: the replay of one set is independent of every other set, and so are node
 * unpacking and index insertion (the reference itself builds its p tables with p threads) ---- */
typedef struct build_job build_job;
typedef void (*job_fn)(build_job *J, int tid);
struct build_job {
	job_fn fn;
	int nthreads;
	/* inputs */
	graph_t *g;
	int nw_keys;
	uint64_t n;
	const uint64_t *keys, *first;
	const uint32_t *l_links, *r_flags, *count;
	/* shared state */
	gnode_t *tmp;
	uint32_t *set_of;
	uint64_t *per_set;       /* p + 1 prefix */
	ord_t *ord;
	uint64_t *hist;          /* nthreads x p */
	volatile int next_set;
};
typedef struct { build_job *J; int tid; } job_arg;

static void *job_thread(void *a)
{
	job_arg *ja = (job_arg *)a;
	ja->J->fn(ja->J, ja->tid);
	return NULL;
}

static void run_parallel(build_job *J, job_fn fn)
{
	J->fn = fn;
	pthread_t th[64];
	job_arg args[64];
	for (int t = 1; t < J->nthreads; t++) {
		args[t].J = J; args[t].tid = t;
		pthread_create(&th[t], NULL, job_thread, &args[t]);
	}
	fn(J, 0);
	for (int t = 1; t < J->nthreads; t++) pthread_join(th[t], NULL);
}

static void job_unpack(build_job *J, int tid)
{
	const uint64_t lo = J->n * (uint64_t)tid / J->nthreads, hi = J->n * (uint64_t)(tid + 1) / J->nthreads;
	const int nwk = J->nw_keys;
	for (uint64_t i = lo; i < hi; i++) {
		gnode_t *nd = &J->tmp[i];
		memset(nd, 0, sizeof *nd);                       /* tmp is malloc'd: first touch happens here, in parallel */
		for (int w = 0; w < nwk; w++) nd->seq.w[4 - nwk + w] = J->keys[i * nwk + w];
		nd->l_links = J->l_links[i];
		nd->r_links = J->r_flags[i] & 0xFFFFFFu;
		nd->linear = (J->r_flags[i] >> 24) & 1; nd->deleted = (J->r_flags[i] >> 25) & 1; nd->single = (J->r_flags[i] >> 27) & 1;
		nd->count = J->count[i];
		J->set_of[i] = (uint32_t)(ref_hash_kmer(&nd->seq, J->g->nw) % (uint64_t)J->g->p);
	}
}

static void job_replay(build_job *J, int tid)
{
	(void)tid;
	graph_t *g = J->g;
	for (;;) {
		const int s = __sync_fetch_and_add(&J->next_set, 1);
		if (s >= g->p) break;
		const uint64_t b = J->per_set[s], e = J->per_set[s + 1];
		const double t_s0 = getenv("SDT_TIMING") ? gb_now() : 0;
		sort_ord(J->ord + b, e - b);
		if (t_s0 > 0 && s == 0) fprintf(stderr, "[graph]      set 0: sort %9.1f ms (%llu nodes)\n", gb_now() - t_s0, (unsigned long long)(e - b));
		uint64_t out = b;                                  /* every node of the set is placed: the set fills [b, e) */
		if (g->nw == 1) {
			replay1_t r;
			memset(&r, 0, sizeof r);
			r.size = next_prime_kh(1024);                 /* init_kmerset(1024, 0.77f), prlHashReads.c:402-423 */
			r.max = (uint64_t)(r.size * 0.77f);
			r.lf = (double)0.77f;
			r.slot = (rslot1 *)malloc(r.size * sizeof(rslot1));
			for (uint64_t i = 0; i < r.size; i++) r.slot[i].id = -1;
			for (uint64_t i = b; i < e; i++) {
				if (i + 24 < e) __builtin_prefetch(&J->tmp[J->ord[i + 24].id].seq.w[3]);
				if (i + 8 < e) __builtin_prefetch(&r.slot[J->tmp[J->ord[i + 8].id].seq.w[3] % r.size], 1);   /* its home slot, if the table does not grow first */
				replay1_put(&r, (int64_t)J->ord[i].id, J->tmp[J->ord[i].id].seq.w[3]);
			}
			if (t_s0 > 0 && s == 0) fprintf(stderr, "[graph]      set 0: sort + puts %9.1f ms\n", gb_now() - t_s0);
			for (uint64_t i = 0; i < r.size; i++) {
				if (i + 8 < r.size && r.slot[i + 8].id >= 0) __builtin_prefetch(&J->tmp[r.slot[i + 8].id]);
				if (r.slot[i].id >= 0) g->nodes[out++] = J->tmp[r.slot[i].id];
			}
			free(r.slot);
			free(r.moved);
			continue;
		}
		replay_t r;
		memset(&r, 0, sizeof r);
		/* init_kmerset(1024, 0.77f), prlHashReads.c:402-423; with -a <n != 0> the 63mer / 127mer binaries ask for
		 * k * 0xFFFFFF slots with k == 0 (:404-413), and init_kmerset turns anything below 3 into 3 (newhash.c:163-166) */
		r.size = graph_init_kmerset_size ? 3 : next_prime_kh(1024);
		r.max = (uint64_t)(r.size * 0.77f);
		r.lf = (double)0.77f;
		r.slot = (int64_t *)malloc(r.size * sizeof(int64_t));
		for (uint64_t i = 0; i < r.size; i++) r.slot[i] = -1;
		for (uint64_t i = b; i < e; i++)
			replay_put(&r, J->tmp, (int64_t)J->ord[i].id, g->nw);
		for (uint64_t i = 0; i < r.size; i++)
			if (r.slot[i] >= 0) g->nodes[out++] = J->tmp[r.slot[i]];
		free(r.slot);
		free(r.moved);
	}
}

static void job_index(build_job *J, int tid)
{
	graph_t *g = J->g;
	const uint64_t lo = g->n * (uint64_t)tid / J->nthreads, hi = g->n * (uint64_t)(tid + 1) / J->nthreads;
	for (uint64_t i = lo; i < hi; i++) {
		uint64_t h = mix_key(&g->nodes[i].seq) & g->index_mask;
		if (g->index64) {
			while (!__sync_bool_compare_and_swap(&g->index64[h], 0ULL, i + 1))
				h = (h + 1) & g->index_mask;
		} else {
			while (!__sync_bool_compare_and_swap(&g->index[h], 0u, (uint32_t)(i + 1)))
				h = (h + 1) & g->index_mask;
		}
	}
}

static void job_count_sets(build_job *J, int tid)
{
	const uint64_t lo = J->n * (uint64_t)tid / J->nthreads, hi = J->n * (uint64_t)(tid + 1) / J->nthreads;
	uint64_t *h = J->hist + (size_t)tid * J->g->p;
	for (uint64_t i = lo; i < hi; i++) h[J->set_of[i]]++;
}

static void job_scatter_sets(build_job *J, int tid)
{
	const uint64_t lo = J->n * (uint64_t)tid / J->nthreads, hi = J->n * (uint64_t)(tid + 1) / J->nthreads;
	uint64_t *h = J->hist + (size_t)tid * J->g->p;
	for (uint64_t i = lo; i < hi; i++) {
		ord_t *o = &J->ord[h[J->set_of[i]]++];
		o->first = J->first[i];
		o->id = i;
	}
}

#define GB_PHASE(name) do { if (getenv("SDT_TIMING")) { double t_ = gb_now(); fprintf(stderr, "[graph]    %-26s %9.1f ms\n", name, t_ - t_sub); t_sub = t_; } } while (0)

static void *free_worker(void *v)
{
	void **p = (void **)v;
	for (int i = 0; i < 4; i++) free(p[i]);
	free(p);
	return NULL;
}

void graph_free_later(void *a, void *b, void *c, void *d)
{
	void **p = (void **)malloc(4 * sizeof(void *));
	pthread_t th;
	if (!p) { free(a); free(b); free(c); free(d); return; }
	p[0] = a; p[1] = b; p[2] = c; p[3] = d;
	if (pthread_create(&th, NULL, free_worker, p) == 0) pthread_detach(th);
	else free_worker(p);
}

int (*graph_index_hook)(graph_t *g, void *user) = NULL;
void *graph_index_hook_user = NULL;

graph_t *graph_build(int K, int nw_variant, int nw_keys, int p, uint64_t n, const uint64_t *keys,
                     const uint32_t *l_links, const uint32_t *r_flags, const uint32_t *count, const uint64_t *first)
{
	double t_sub = gb_now();
	graph_t *g = (graph_t *)calloc(1, sizeof *g);
	g->K = K; g->nw = nw_variant; g->p = p; g->n = n;
	build_job J;
	memset(&J, 0, sizeof J);
	J.nthreads = par_threads();
	J.g = g; J.nw_keys = nw_keys; J.n = n; J.keys = keys; J.first = first;
	J.l_links = l_links; J.r_flags = r_flags; J.count = count;
	J.tmp = (gnode_t *)malloc((n ? n : 1) * sizeof(gnode_t));
	J.set_of = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
	J.per_set = (uint64_t *)calloc((size_t)p + 1, sizeof(uint64_t));
	run_parallel(&J, job_unpack);
	GB_PHASE("unpack + hash_kmer");
	/* group node ids by set (counting sort, per-thread histograms); each group is then ordered by first
	 * occurrence and replayed by one worker */
	J.ord = (ord_t *)malloc((n ? n : 1) * sizeof(ord_t));
	J.hist = (uint64_t *)calloc((size_t)J.nthreads * (size_t)p + 1, sizeof(uint64_t));
	run_parallel(&J, job_count_sets);
	{
		uint64_t acc = 0;
		for (int sidx = 0; sidx < p; sidx++) {
			J.per_set[sidx] = acc;
			for (int t = 0; t < J.nthreads; t++) {
				const uint64_t c = J.hist[(size_t)t * p + sidx];
				J.hist[(size_t)t * p + sidx] = acc;           /* where thread t starts writing in set sidx */
				acc += c;
			}
		}
		J.per_set[p] = acc;
	}
	run_parallel(&J, job_scatter_sets);
	free(J.hist);
	GB_PHASE("group by set");
	g->nodes = (gnode_t *)malloc((n ? n : 1) * sizeof(gnode_t));      /* every element is written by job_replay */
	g->set_start = (uint64_t *)calloc((size_t)p + 1, sizeof(uint64_t));
	memcpy(g->set_start, J.per_set, ((size_t)p + 1) * sizeof(uint64_t));
	J.next_set = 0;
	run_parallel(&J, job_replay);
	GB_PHASE("sort + replay per set");
	/* gigabytes of scratch: returning them to the kernel takes a fraction of a second, off the critical path */
	graph_free_later(J.ord, J.per_set, J.set_of, J.tmp);
	/* index */
	/* node ids past 32 bits: 64-bit index entries, built here (the device-built index is the 32-bit one; the hook still
	 * brings the device mirror of the graph up, then declines) */
	const int wide = n >= 0xFFFFFFFEULL || getenv("SDT_WIDE_INDEX") != NULL;
	uint64_t cap = 1024;
	while (cap < 2 * n + 2) cap <<= 1;
	if (wide) {
		g->index64 = (uint64_t *)calloc(cap, sizeof(uint64_t));
		if (!g->index64) { printf("out of memory for the node index (%llu entries)\n", (unsigned long long)cap); exit(1); }
		g->index_mask = cap - 1;
	}
	if (graph_index_hook && graph_index_hook(g, graph_index_hook_user) == 0) {
		GB_PHASE("index (device)");
		return g;
	}
	if (!wide) {
		g->index = (uint32_t *)calloc(cap, sizeof(uint32_t));
		g->index_mask = cap - 1;
	}
	run_parallel(&J, job_index);
	GB_PHASE("index");
	return g;
}

void graph_free(graph_t *g)
{
	if (!g) return;
	free(g->nodes); free(g->set_start); free(g->index); free(g->index64); free(g->patch); free(g->tlist); free(g->dirty); free(g->dlist); free(g);
}
