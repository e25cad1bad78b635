/* graph.c -- see graph.h: layout replay, flat node array, lookup index, hash_kmer. */
#include "../../sdt_knobs.h"
#include "graph.h"
#include <pthread.h>
#include <math.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include "par.h"
#include "big.h"        /* every large block (replay tables, node arrays) asks for transparent huge pages */
#include <time.h>
static double gb_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

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
int graph_force_wide_index = 0;
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

/* One table for the whole replay of a set, as the reference has one array that it reallocs: an entry is (key, id, tag) with
 * tag 0 = empty, tag == gen = placed in the current geometry, tag == gen - 1 (during a rehash) = still where the previous
 * geometry left it -- "empty" for the probing of the new geometry, evicted and carried onward when something lands on it.  A
 * step of the rehash then touches ONE random cache line (the landing slot holds flags, key and id together), the scan over the
 * old slots is sequential, and the homes of the entries a few steps ahead are prefetched.  The table is allocated once at its
 * final size (the sequence of sizes depends only on the number of keys). */
typedef struct { uint64_t key; uint32_t id, tag; } rent_t;         /* 1-word keys: the key travels with the entry */
typedef struct { uint32_t id, tag; } rent_w;                       /* wider keys: looked up through the id */

static uint64_t replay_next_size(uint64_t size, double lf, uint64_t count)
{
	uint64_t n = size;
	do {
		n = n < 0xFFFFFFFu ? n << 1 : n + 0xFFFFFFu;
		n = next_prime_kh(n);
	} while (n * lf < (double)(count + 1));
	return n;
}

static uint64_t replay_final_size(uint64_t init, uint64_t m)
{
	uint64_t size = init, max = (uint64_t)(size * 0.77f);
	const double lf = (double)0.77f;
	while (m > max) {                                       /* a put grows the table when count + 1 > max */
		size = replay_next_size(size, lf, max);             /* ... and that happens at count == max */
		max = (uint64_t)(size * lf);
	}
	return size;
}

#ifndef RP_AHEAD
#define RP_AHEAD 16
#endif
#ifndef RP_PF
#define RP_PF(p) __builtin_prefetch((p), 1)
#endif

/* keys[i] (i = 0..m-1, first-occurrence order) -> ids in slot order */
static void replay_set1(const uint64_t *keys, uint64_t m, uint64_t init, uint64_t base, uint64_t *out)
{
	if (m > 0xFFFFFFF0ULL) { printf("a set of %llu nodes does not fit the replay's 32-bit ids\n", (unsigned long long)m); exit(1); }
	const uint64_t fin = replay_final_size(init, m);
	rent_t *t = (rent_t *)malloc(fin * sizeof(rent_t));
	if (!t) { printf("out of memory for a replay table of %llu slots\n", (unsigned long long)fin); exit(1); }
	memset(t, 0, fin * sizeof(rent_t));
	uint64_t size = init, count = 0, max = (uint64_t)(size * 0.77f);
	const double lf = (double)0.77f;
	uint32_t gen = 1;
	uint64_t hring[RP_AHEAD], ring_size = 0;
	for (uint64_t i = 0; i < m; i++) {
		if (count + 1 > max) {
			/* encap_kmerset (newhash.c:293-409) */
			const uint64_t old = size, n = replay_next_size(size, lf, count);
			const uint32_t was = gen++;
			uint64_t ring[RP_AHEAD];
			for (uint64_t j = 0; j < RP_AHEAD && j < old; j++) {
				ring[j] = t[j].tag == was ? t[j].key % n : 0;
				RP_PF(&t[ring[j]]);
			}
			for (uint64_t j = 0; j < old; j++) {
				const uint64_t home = ring[j % RP_AHEAD];
				if (j + RP_AHEAD < old) {
					const uint64_t hp = t[j + RP_AHEAD].tag == was ? t[j + RP_AHEAD].key % n : 0;
					ring[j % RP_AHEAD] = hp;
					RP_PF(&t[hp]);
				}
#ifdef RP_SECOND
				{   /* the landing slot of the entry half way ahead is in cache by now: an unmoved entry sitting there will be evicted
				     * and carried to ITS home -- ask for that line as well */
					const uint64_t hq = ring[(j + RP_AHEAD / 2) % RP_AHEAD];
					if (hq < old && hq > j && t[hq].tag == was) RP_PF(&t[t[hq].key % n]);
				}
#endif
				if (t[j].tag != was) continue;               /* empty, or evicted earlier in this rehash */
				rent_t carry = t[j];
				t[j].tag = 0;
				uint64_t h = home;
				for (;;) {
					while (t[h].tag == gen) h = h + 1 == n ? 0 : h + 1;
					if (h < old && t[h].tag == was) {        /* an entry that has not moved yet: it gives way and is carried on */
						const rent_t evicted = t[h];
						t[h] = carry;
						t[h].tag = gen;
						carry = evicted;
						h = carry.key % n;
						continue;
					}
					t[h] = carry;
					t[h].tag = gen;
					break;
				}
			}
			size = n;
			max = (uint64_t)(n * lf);
		}
		/* the homes of the keys ahead are computed once, prefetched, and kept in a ring (a growth in between refills it) */
		if (i == 0 || count == 0 || ring_size != size) {
			for (uint64_t j = 0; j < RP_AHEAD && i + j < m; j++) { hring[(i + j) % RP_AHEAD] = keys[i + j] % size; RP_PF(&t[hring[(i + j) % RP_AHEAD]]); }
			ring_size = size;
		}
		uint64_t h = hring[i % RP_AHEAD];
		if (i + RP_AHEAD < m) { const uint64_t hp = keys[i + RP_AHEAD] % size; hring[i % RP_AHEAD] = hp; RP_PF(&t[hp]); }
		while (t[h].tag) h = h + 1 == size ? 0 : h + 1;
		t[h].key = keys[i];
		t[h].id = (uint32_t)i;
		t[h].tag = gen;
		count++;
	}
	uint64_t k = 0;
	for (uint64_t s = 0; s < size; s++)
		if (t[s].tag) out[k++] = base + t[s].id;
	free(t);
}

static uint64_t home_words(const uint64_t *k, int nwk, uint64_t size)
{
	if (nwk == 1) return k[0] % size;                                    /* newhash.c:428 (and :423-425, :43-55 with zero high words) */
	if (nwk == 2) {                                                      /* :423-425 */
		unsigned __int128 v = ((unsigned __int128)k[0] << 64) | k[1];
		return (uint64_t)(v % size);
	}
	uint64_t t = k[0] % size;                                            /* :43-55, 32 bits at a time */
	const uint64_t part[6] = {k[1] >> 32, k[1] & 0xffffffffu, k[2] >> 32, k[2] & 0xffffffffu, k[3] >> 32, k[3] & 0xffffffffu};
	for (int i = 0; i < 6; i++)
		t = ((t << 32) | part[i]) % size;
	return t;
}

static void replay_setw(const uint64_t *keys, int nwk, uint64_t m, uint64_t init, uint64_t base, uint64_t *out)
{
	if (m > 0xFFFFFFF0ULL) { printf("a set of %llu nodes does not fit the replay's 32-bit ids\n", (unsigned long long)m); exit(1); }
	const uint64_t fin = replay_final_size(init, m);
	rent_w *t = (rent_w *)malloc(fin * sizeof(rent_w));
	if (!t) { printf("out of memory for a replay table of %llu slots\n", (unsigned long long)fin); exit(1); }
	memset(t, 0, fin * sizeof(rent_w));
	uint64_t size = init, count = 0, max = (uint64_t)(size * 0.77f);
	const double lf = (double)0.77f;
	uint32_t gen = 1;
	for (uint64_t i = 0; i < m; i++) {
		if (count + 1 > max) {
			const uint64_t old = size, n = replay_next_size(size, lf, count);
			const uint32_t was = gen++;
			for (uint64_t j = 0; j < old; j++) {
				if (j + 8 < old && t[j + 8].tag == was) __builtin_prefetch(&keys[(uint64_t)t[j + 8].id * nwk]);
				if (t[j].tag != was) continue;
				uint32_t carry = t[j].id;
				t[j].tag = 0;
				for (;;) {
					uint64_t h = home_words(keys + (uint64_t)carry * nwk, nwk, n);
					while (t[h].tag == gen) h = h + 1 == n ? 0 : h + 1;
					if (h < old && t[h].tag == was) {
						const uint32_t evicted = t[h].id;
						t[h].id = carry;
						t[h].tag = gen;
						carry = evicted;
						continue;
					}
					t[h].id = carry;
					t[h].tag = gen;
					break;
				}
			}
			size = n;
			max = (uint64_t)(n * lf);
		}
		uint64_t h = home_words(keys + i * nwk, nwk, size);
		while (t[h].tag) h = h + 1 == size ? 0 : h + 1;
		t[h].id = (uint32_t)i;
		t[h].tag = gen;
		count++;
	}
	uint64_t k = 0;
	for (uint64_t s = 0; s < size; s++)
		if (t[s].tag) out[k++] = base + t[s].id;
	free(t);
}

/* init_kmerset(1024, 0.77f), prlHashReads.c:402-423; with -a <n != 0> the 63mer / 127mer binaries ask for k * 0xFFFFFF slots
 * with k == 0 (:404-413), and init_kmerset turns anything below 3 into 3 (newhash.c:163-166) */
static uint64_t replay_init_size(int nw_variant)
{
	return nw_variant != 1 && graph_init_kmerset_size ? 3 : next_prime_kh(1024);
}

typedef struct { int nw_variant, nwk, p; const uint64_t *keys, *set_start; uint64_t *order; volatile int next_set; } rp_job;

static void *rp_thread(void *v)
{
	rp_job *J = (rp_job *)v;
	for (;;) {
		const int s = __sync_fetch_and_add(&J->next_set, 1);
		if (s >= J->p) break;
		const uint64_t b = J->set_start[s], m = J->set_start[s + 1] - b;
		const double t0 = sdt_env("SDT_TIMING") && s == 0 ? gb_now() : 0;
		if (J->nwk == 1) replay_set1(J->keys + b, m, replay_init_size(J->nw_variant), b, J->order + b);
		else replay_setw(J->keys + b * J->nwk, J->nwk, m, replay_init_size(J->nw_variant), b, J->order + b);
		if (t0 > 0) {
			fprintf(stderr, "[graph]      set 0: replay %9.1f ms (%llu nodes)\n", gb_now() - t0, (unsigned long long)m);
			FILE *f = fopen("/proc/self/smaps_rollup", "r");              /* are the big tables on huge pages? */
			char line[256];
			while (f && fgets(line, sizeof line, f))
				if (!strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Rss:", 4)) fprintf(stderr, "[graph]      %s", line);
			if (f) fclose(f);
		}
	}
	return NULL;
}

/* keys grouped by set ([set_start[s], set_start[s + 1])) and, inside a set, in first-occurrence order (nwk words each, most
 * significant first): order[v] = index into keys[] of the node at visiting position v.  Sets are replayed side by side. */
void graph_replay_order(int nw_variant, int nwk, int p, const uint64_t *keys, const uint64_t *set_start, uint64_t *order)
{
	rp_job J = {nw_variant, nwk, p, keys, set_start, order, 0};
	int nt = par_threads();
	if (nt > p) nt = p;
	pthread_t th[64];
	for (int t = 1; t < nt; t++) pthread_create(&th[t], NULL, rp_thread, &J);
	rp_thread(&J);
	for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
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
	uint64_t *skeys, *order; /* keys grouped by set in first-occurrence order; visiting position -> index into them */
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

/* per set: sort by first occurrence, then the keys in that order (what the device hands over ready-made) */
static void job_sort_sets(build_job *J, int tid)
{
	(void)tid;
	graph_t *g = J->g;
	const int nwk = J->nw_keys;
	for (;;) {
		const int s = __sync_fetch_and_add(&J->next_set, 1);
		if (s >= g->p) break;
		const uint64_t b = J->per_set[s], e = J->per_set[s + 1];
		sort_ord(J->ord + b, e - b);
		for (uint64_t i = b; i < e; i++)
			for (int w = 0; w < nwk; w++) J->skeys[i * nwk + w] = J->keys[J->ord[i].id * nwk + w];
	}
}

static void job_place(build_job *J, int tid)
{
	const uint64_t lo = J->n * (uint64_t)tid / J->nthreads, hi = J->n * (uint64_t)(tid + 1) / J->nthreads;
	for (uint64_t v = lo; v < hi; v++) {
		if (v + 8 < hi) __builtin_prefetch(&J->tmp[J->ord[J->order[v + 8]].id]);
		J->g->nodes[v] = J->tmp[J->ord[J->order[v]].id];
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

#define GB_PHASE(name) do { if (sdt_env("SDT_TIMING")) { double t_ = gb_now(); fprintf(stderr, "[graph]    %-26s %9.1f ms\n", name, t_ - t_sub); t_sub = t_; } } while (0)

/* free() of a multi-gigabyte block is one munmap that holds the address-space lock for its whole length, and every page fault
 * of every other thread waits behind it (a 32 GB node array let go beside the second read pass stalled the arcs' download for
 * 1.5 s).  Large blocks give their pages back first, in pieces, under the shared lock (MADV_DONTNEED); the munmap that follows
 * finds nothing left to do. */
#include <sys/mman.h>
#include <malloc.h>
typedef struct { char *a, *e; } rp_part;
static void *release_part(void *v)
{
	rp_part *P = (rp_part *)v;
	const size_t piece = (size_t)64 << 20;
	for (char *a = P->a; a < P->e; a += piece)
		madvise(a, (size_t)(P->e - a) < piece ? (size_t)(P->e - a) : piece, MADV_DONTNEED);
	return NULL;
}

/* (the boxes' kernels clear pages when they are freed, 25-40 ms per GiB and thread: a 32 GB node array takes one thread more than a
 * second, which the process then spends in exit; eight threads give their shares back side by side) */
static void release_pages(void *p)
{
	if (!p) return;
	const size_t sz = malloc_usable_size(p), page = 4096;
	if (sz < ((size_t)1 << 30)) return;
	char *a = (char *)(((uintptr_t)p + page - 1) & ~(uintptr_t)(page - 1)), *e = (char *)(((uintptr_t)p + sz) & ~(uintptr_t)(page - 1));
	enum { NT = 8 };
	pthread_t th[NT];
	rp_part part[NT];
	const size_t share = (((size_t)(e - a) / NT) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
	int started[NT];
	for (int t = 0; t < NT; t++) {
		part[t].a = a + share * (size_t)t < e ? a + share * (size_t)t : e;
		part[t].e = a + share * (size_t)(t + 1) < e ? a + share * (size_t)(t + 1) : e;
		started[t] = part[t].a < part[t].e && pthread_create(&th[t], NULL, release_part, &part[t]) == 0;
		if (!started[t]) release_part(&part[t]);
	}
	for (int t = 0; t < NT; t++) if (started[t]) pthread_join(th[t], NULL);
}

static void *free_worker(void *v)
{
	void **p = (void **)v;
	for (int i = 0; i < 4; i++) { release_pages(p[i]); free(p[i]); }
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
	J.skeys = (uint64_t *)malloc((n ? n : 1) * (size_t)nw_keys * sizeof(uint64_t));
	J.order = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
	run_parallel(&J, job_sort_sets);
	GB_PHASE("sort per set");
	graph_replay_order(nw_variant, nw_keys, p, J.skeys, J.per_set, J.order);
	GB_PHASE("replay per set");
	run_parallel(&J, job_place);
	GB_PHASE("nodes into visiting order");
	free(J.skeys); free(J.order);
	/* gigabytes of scratch: returning them to the kernel takes a fraction of a second, off the critical path */
	graph_free_later(J.ord, J.per_set, J.set_of, J.tmp);
	/* index */
	/* node ids past 32 bits: 64-bit index entries, built here (the device-built index is the 32-bit one; the hook still
	 * brings the device mirror of the graph up, then declines) */
	const int wide = n >= 0xFFFFFFFEULL || graph_force_wide_index || sdt_test_env("SDT_WIDE_INDEX") != NULL;
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

/* the graph from nodes that are in visiting order already (sdt_gpu_export_ordered after graph_replay_order); the look-up index
 * comes from graph_index_hook (the device builds it) or is built here */
typedef struct { graph_t *g; int nwk; const uint64_t *keys; const uint32_t *l_links, *r_flags, *count; } fo_ctx;

static void fo_unpack(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	fo_ctx *F = (fo_ctx *)vc;
	const int nwk = F->nwk;
	for (uint64_t i = lo; i < hi; i++) {
		gnode_t *nd = &F->g->nodes[i];
		memset(nd, 0, sizeof *nd);
		for (int w = 0; w < nwk; w++) nd->seq.w[4 - nwk + w] = F->keys[i * nwk + w];
		nd->l_links = F->l_links[i];
		nd->r_links = F->r_flags[i] & 0xFFFFFFu;
		nd->linear = (F->r_flags[i] >> 24) & 1; nd->deleted = (F->r_flags[i] >> 25) & 1; nd->single = (F->r_flags[i] >> 27) & 1;
		nd->count = F->count[i];
	}
}

static void fo_index(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	graph_t *g = ((fo_ctx *)vc)->g;
	for (uint64_t i = lo; i < hi; i++) {
		uint64_t h = mix_key(&g->nodes[i].seq) & g->index_mask;
		if (g->index64) { while (!__sync_bool_compare_and_swap(&g->index64[h], 0ULL, i + 1)) h = (h + 1) & g->index_mask; }
		else { while (!__sync_bool_compare_and_swap(&g->index[h], 0u, (uint32_t)(i + 1))) h = (h + 1) & g->index_mask; }
	}
}

int graph_index_hook_early = 0;
typedef struct { graph_t *g; int rc; } ih_job;
static void *index_hook_thread(void *v)
{
	ih_job *J = (ih_job *)v;
	J->rc = graph_index_hook(J->g, graph_index_hook_user);
	return NULL;
}

graph_t *graph_from_ordered(int K, int nw_variant, int nw_keys, int p, uint64_t n, const uint64_t *keys, const uint32_t *l_links,
                            const uint32_t *r_flags, const uint32_t *count, const uint64_t *set_start)
{
	double t_sub = gb_now();
	graph_t *g = (graph_t *)calloc(1, sizeof *g);
	g->K = K; g->nw = nw_variant; g->p = p; g->n = n;
	g->nodes = (gnode_t *)malloc((n ? n : 1) * sizeof(gnode_t));
	g->set_start = (uint64_t *)calloc((size_t)p + 1, sizeof(uint64_t));
	memcpy(g->set_start, set_start, ((size_t)p + 1) * sizeof(uint64_t));
	fo_ctx F = {g, nw_keys, keys, l_links, r_flags, count};
	const int wide = n >= 0xFFFFFFFEULL || graph_force_wide_index || sdt_test_env("SDT_WIDE_INDEX") != NULL;
	uint64_t cap = 1024;
	while (cap < 2 * n + 2) cap <<= 1;
	if (wide) {
		g->index64 = (uint64_t *)calloc(cap, sizeof(uint64_t));
		if (!g->index64) { printf("out of memory for the node index (%llu entries)\n", (unsigned long long)cap); exit(1); }
		g->index_mask = cap - 1;
	}
	/* the device builds the index from ITS copy of the node order and sends it over while the host's threads unpack the nodes
	 * (graph_index_hook_early: the hook reads nothing of nodes[]) */
	ih_job IH = {g, 1};
	pthread_t ih;
	const int early = graph_index_hook && graph_index_hook_early && pthread_create(&ih, NULL, index_hook_thread, &IH) == 0;
	par_for(0, n, 1 << 16, fo_unpack, &F);
	GB_PHASE("unpack");
	if (early) pthread_join(ih, NULL);
	else if (graph_index_hook) IH.rc = graph_index_hook(g, graph_index_hook_user);
	if (graph_index_hook && IH.rc == 0) {
		GB_PHASE(early ? "index (device, beside the unpacking): the rest" : "index (device)");
		return g;
	}
	if (!wide) {
		g->index = (uint32_t *)calloc(cap, sizeof(uint32_t));
		g->index_mask = cap - 1;
	}
	par_for(0, n, 1 << 14, fo_index, &F);
	GB_PHASE("index");
	return g;
}

static void clear_dirty_part(void *vc, uint64_t lo, uint64_t hi, int tid)
{
	(void)tid;
	graph_t *g = (graph_t *)vc;
	for (uint64_t k = lo; k < hi; k++) g->dirty[g->dlist[k]] = 0;
}

void graph_clear_dirty(graph_t *g)
{
	if (g->dn < 65536) { for (size_t k = 0; k < g->dn; k++) g->dirty[g->dlist[k]] = 0; return; }
	par_for(0, g->dn, 1 << 16, clear_dirty_part, g);
}

void graph_free(graph_t *g)
{
	if (!g) return;
	free(g->nodes); free(g->set_start); free(g->index); free(g->index64); free(g->patch); free(g->tlist); free(g->dirty); free(g->dlist); free(g);
}
