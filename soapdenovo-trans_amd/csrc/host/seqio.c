/* seqio.c -- see seqio.h */
#define _GNU_SOURCE
#include "../sdt_knobs.h"
#include "seqio.h"
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct {
	const char *beg, *end;      /* [beg, end) holds whole records */
	sdt_batch out;
	int done;
	int owner, encode;          /* sdt_read_shard_begin */
	uint64_t index;             /* number of the chunk in the pass */
	int slot;                   /* pool buffer taken with the chunk number, -1 = none */
} chunk_t;

static int g_shard_rank = 0, g_shard_n = 1, g_shard_keep_all = 0, g_shard_skip = 0;
static uint64_t g_shard_chunk = 0;
uint64_t sdt_reader_bytes_parsed = 0, sdt_reader_bytes_seen = 0;

void sdt_read_shard_begin(int rank, int nranks, int keep_all)
{
	g_shard_rank = rank;
	g_shard_n = nranks > 0 ? nranks : 1;
	g_shard_keep_all = keep_all;
	g_shard_chunk = 0;
	g_shard_skip = 0;
	sdt_reader_bytes_parsed = sdt_reader_bytes_seen = 0;
}

void sdt_read_shard_skip_foreign(int on) { g_shard_skip = on; }

typedef struct {
	chunk_t *chunks;
	int nchunks, next, turn;    /* next: chunk numbers handed out; turn: chunks whose pool buffer has been handed out */
	int fmt, max_read_len, reverse;
	pthread_mutex_t mu;
	pthread_cond_t cv;
} job_t;

#include <time.h>
double sdt_reader_wait_ms = 0, sdt_reader_fn_ms = 0;      /* consumer thread: waiting for the parsers / inside the callback */
static double seq_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

/* ---- pool of output buffers (seqio.h) ---- */
typedef struct { uint32_t *w; uint64_t *o; size_t wcap, ocap; int state; } pslot_t;      /* state: 0 free, 1 in use, 2 taken by the consumer */
static struct {
	void *(*alloc)(size_t);
	void (*release)(void *);
	pslot_t *slot;
	int n, on;
	pthread_mutex_t mu;
	pthread_cond_t cv;
} g_pool = {NULL, NULL, NULL, 0, 0, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER};

void sdt_pool_enable(void *(*alloc)(size_t bytes), void (*release)(void *p), int nslots)
{
	sdt_pool_disable();
	g_pool.alloc = alloc; g_pool.release = release;
	g_pool.slot = (pslot_t *)calloc((size_t)nslots, sizeof(pslot_t));
	g_pool.n = nslots;
	g_pool.on = 1;
}

void sdt_pool_disable(void)
{
	for (int i = 0; i < g_pool.n; i++) {
		if (g_pool.slot[i].w) g_pool.release(g_pool.slot[i].w);
		if (g_pool.slot[i].o) g_pool.release(g_pool.slot[i].o);
	}
	free(g_pool.slot);
	g_pool.slot = NULL;
	g_pool.n = g_pool.on = 0;
}

void sdt_pool_take(int slot)
{
	pthread_mutex_lock(&g_pool.mu);
	if (slot >= 0 && slot < g_pool.n) g_pool.slot[slot].state = 2;
	pthread_mutex_unlock(&g_pool.mu);
}

void sdt_pool_release(int slot)
{
	pthread_mutex_lock(&g_pool.mu);
	if (slot >= 0 && slot < g_pool.n) g_pool.slot[slot].state = 0;
	pthread_cond_broadcast(&g_pool.cv);
	pthread_mutex_unlock(&g_pool.mu);
}

/* a free slot (blocks until there is one); its buffers hold `text` bytes of input: 2 bits per byte and one offset per 64 bytes
 * (a chunk with shorter records falls back to malloc) */
static int pool_acquire_locked(pthread_mutex_t *held, size_t text)
{
	(void)held;
	for (;;) {
		for (int i = 0; i < g_pool.n; i++)
			if (g_pool.slot[i].state == 0) {
				pslot_t *s = &g_pool.slot[i];
				s->state = 1;
				const size_t wneed = text / 16 + 64, oneed = text / 64 + 64;
				if (s->wcap < wneed) {
					if (s->w) g_pool.release(s->w);
					s->w = (uint32_t *)g_pool.alloc(wneed * sizeof(uint32_t));
					s->wcap = s->w ? wneed : 0;
				}
				if (s->ocap < oneed) {
					if (s->o) g_pool.release(s->o);
					s->o = (uint64_t *)g_pool.alloc(oneed * sizeof(uint64_t));
					s->ocap = s->o ? oneed : 0;
				}
				if (!s->w || !s->o) { s->state = 0; return -1; }          /* no pinned memory: this chunk goes the malloc way */
				return i;
			}
		pthread_cond_wait(&g_pool.cv, &g_pool.mu);
	}
}

/* ---- 2-bit stream writer ---- */
typedef struct {
	uint32_t *w;
	uint64_t cap, nbases;
	int fixed;                  /* w is a pool buffer of `cap` words: it cannot grow (the chunk then moves to a malloc block) */
} packer_t;

static void pk_reserve(packer_t *p, uint64_t more_bases)
{
	uint64_t need = ((p->nbases + more_bases + 15) >> 4) + 8;
	if (need > p->cap) {
		uint64_t ncap = p->cap ? p->cap * 2 : 1 << 16;
		while (ncap < need) ncap *= 2;
		if (p->fixed) {                                   /* (cannot happen for buffers sized by the chunk's text: kept for safety) */
			uint32_t *nw = (uint32_t *)malloc(ncap * sizeof(uint32_t));
			memcpy(nw, p->w, p->cap * sizeof(uint32_t));
			p->w = nw;
			p->fixed = 0;
		} else {
			p->w = (uint32_t *)realloc(p->w, ncap * sizeof(uint32_t));
		}
		memset(p->w + p->cap, 0, (ncap - p->cap) * sizeof(uint32_t));
		p->cap = ncap;
	}
}

static inline void pk_put(packer_t *p, unsigned code)
{
	p->w[p->nbases >> 4] |= (uint32_t)code << (30 - 2 * (p->nbases & 15));
	p->nbases++;
}

/* Fast path for the common line -- nothing but letters: 32 characters per step.  The base code is bits 1..2 of the
 * character in either case ((c & 6) >> 1, inc/def.h:39-42), so after a check that all 32 bytes are letters the codes
 * are one shift and one mask away, and PEXT squeezes the 2-bit codes of 8 bytes into 16 bits (first base on top).
 * Returns how many leading characters it consumed (all of them unless a block holds a non-letter); the caller's
 * scalar loop does the rest. */
#include <immintrin.h>
__attribute__((target("avx2,bmi2")))
static inline int letters32(const char *p32, uint32_t g[2])
{
	const __m256i lower = _mm256_set1_epi8(0x20), a = _mm256_set1_epi8('a'), z = _mm256_set1_epi8(25), three = _mm256_set1_epi8(3);
	const __m256i c = _mm256_loadu_si256((const __m256i *)p32);
	const __m256i t = _mm256_sub_epi8(_mm256_or_si256(c, lower), a);               /* letter <=> 0..25 (unsigned) */
	if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_min_epu8(t, z), t)) != -1) return 0;
	const __m256i codes = _mm256_and_si256(_mm256_srli_epi16(c, 1), three);
	uint64_t lane[4];
	_mm256_storeu_si256((__m256i *)lane, codes);
	for (int h = 0; h < 2; h++) {
		const uint64_t hi = _pext_u64(__builtin_bswap64(lane[2 * h]), 0x0303030303030303ULL);
		const uint64_t lo = _pext_u64(__builtin_bswap64(lane[2 * h + 1]), 0x0303030303030303ULL);
		g[h] = (uint32_t)((hi << 16) | lo);
	}
	return 1;
}

/* OR 32 bits (16 bases, first base on top) into the stream at base position pos; words past the stream end are zero */
static inline void append16(packer_t *pk, uint64_t pos, uint32_t g)
{
	const uint64_t wi = pos >> 4;
	const int bo = 2 * (int)(pos & 15);
	const uint64_t v = (uint64_t)g << (32 - bo);
	pk->w[wi] |= (uint32_t)(v >> 32);
	pk->w[wi + 1] |= (uint32_t)v;
}

__attribute__((target("avx2,bmi2")))
static int encode_letters_avx2(packer_t *pk, const char *s, int n)
{
	int i = 0;
	uint32_t g[2];
	for (; i + 32 <= n; i += 32) {
		if (!letters32(s + i, g)) return i;
		append16(pk, pk->nbases, g[0]);
		append16(pk, pk->nbases + 16, g[1]);
		pk->nbases += 32;
	}
	if (i < n) {
		/* the last 1..31 characters, padded with 'A' (code 0: OR-ing its bits changes nothing) */
		char buf[32];
		const int r = n - i;
		memcpy(buf, s + i, (size_t)r);
		memset(buf + r, 'A', (size_t)(32 - r));
		if (!letters32(buf, g)) return i;
		append16(pk, pk->nbases, g[0]);
		append16(pk, pk->nbases + 16, g[1]);
		pk->nbases += (uint64_t)r;
		i = n;
	}
	return i;
}

static int have_avx2_bmi2(void)
{
	static int cached = -1;                              /* parser threads race to fill it: relaxed atomics, same value */
	int c = __atomic_load_n(&cached, __ATOMIC_RELAXED);
	if (c < 0) {
		c = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !sdt_test_env("SDT_NO_SIMD");
		__atomic_store_n(&cached, c, __ATOMIC_RELAXED);
	}
	return c;
}

/* one sequence line -> stream; returns coded length */
static int encode_line(packer_t *pk, const char *s, int n, int max_read_len, int reverse)
{
	unsigned char tmp[8192];
	unsigned char *codes = tmp;
	if (n > max_read_len) n = max_read_len;
	int fast = 0;
	if (!reverse && n >= 8 && have_avx2_bmi2()) {
		pk_reserve(pk, (uint64_t)n);
		fast = encode_letters_avx2(pk, s, n);
		s += fast;
		n -= fast;
	}
	if (n > (int)sizeof tmp) codes = (unsigned char *)malloc((size_t)n);
	int m = 0;
	for (int i = 0; i < n; i++) {
		unsigned char c = (unsigned char)s[i];
		if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 'a' + 'A');
		if (c >= 'A' && c <= 'Z') codes[m++] = (unsigned char)((c & 6) >> 1);
		else if (c == '.') codes[m++] = 0;
	}
	pk_reserve(pk, (uint64_t)m);
	if (!reverse)
		for (int i = 0; i < m; i++) pk_put(pk, codes[i]);
	else
		for (int i = m - 1; i >= 0; i--) pk_put(pk, codes[i] ^ 2u);
	if (codes != tmp) free(codes);
	return m + fast;
}

static inline const char *line_end(const char *p, const char *end)
{
	const char *q = (const char *)memchr(p, '\n', (size_t)(end - p));
	return q ? q : end;
}

static void parse_chunk(job_t *J, chunk_t *c)
{
	if (!c->encode && g_shard_skip) {                      /* a foreign chunk nobody here needs to count: not a byte of it is read */
		memset(&c->out, 0, sizeof c->out);
		c->out.pool_slot = -1;
		c->out.owner = c->owner;
		c->out.counted_only = 1;
		c->out.count_unknown = 1;
		c->out.text_bytes = (uint64_t)(c->end - c->beg);
		c->out.chunk_index = c->index;
		return;
	}
	packer_t pk = {0};
	uint64_t cap_off = 1024, n = 0;
	uint64_t *offs;
	int offs_pooled = 0;
	if (c->slot >= 0 && c->encode) {
		pslot_t *ps = &g_pool.slot[c->slot];
		pk.w = ps->w; pk.cap = ps->wcap; pk.fixed = 1;
		/* only the words this chunk can reach need to be zero (the packer ORs bases in) */
		const size_t reach = (size_t)(c->end - c->beg) / 16 + 64;
		memset(pk.w, 0, (reach < ps->wcap ? reach : ps->wcap) * sizeof(uint32_t));
		offs = ps->o; cap_off = ps->ocap; offs_pooled = 1;
	} else {
		offs = (uint64_t *)malloc(cap_off * sizeof(uint64_t));
	}
	offs[0] = 0;
	uint64_t first_len = 0;
	int same_len = 1;
	const char *p = c->beg, *end = c->end;
	pk_reserve(&pk, 0);
	while (p < end) {
		const char *e = line_end(p, end);             /* header line */
		if (e == p) { p = e + 1; continue; }          /* blank line between records */
		const char *seq = e < end ? e + 1 : end;
		int len = 0;
		if (J->fmt == 'q') {
			const char *se = line_end(seq, end);
			int sl = (int)(se - seq);
			if (sl > 0 && seq[sl - 1] == '\r') sl--;
			if (c->encode) len = encode_line(&pk, seq, sl, J->max_read_len, J->reverse);
			const char *plus = se < end ? se + 1 : end;
			const char *pe = line_end(plus, end);
			const char *qual = pe < end ? pe + 1 : end;
			const char *qe = line_end(qual, end);
			p = qe < end ? qe + 1 : end;
		} else {
			/* FASTA: sequence lines up to the next '>' at a line start are one read (the reference only
			 * handles the one-line form correctly; multi-line records are concatenated here) */
			char stackbuf[16384];
			char *buf = stackbuf;
			size_t bl = 0, bcap = sizeof stackbuf;
			const char *q = seq;
			while (q < end && *q != '>') {
				const char *se = line_end(q, end);
				size_t sl = (size_t)(se - q);
				if (sl > 0 && q[sl - 1] == '\r') sl--;
				if (!c->encode) sl = 0;
				if (bl + sl > bcap) {
					bcap = (bl + sl) * 2;
					char *nb = (char *)malloc(bcap);
					memcpy(nb, buf, bl);
					if (buf != stackbuf) free(buf);
					buf = nb;
				}
				memcpy(buf + bl, q, sl);
				bl += sl;
				q = se < end ? se + 1 : end;
			}
			if (c->encode) len = encode_line(&pk, buf, (int)bl, J->max_read_len, J->reverse);
			if (buf != stackbuf) free(buf);
			p = q;
		}
		if (n + 2 > cap_off) {
			cap_off *= 2;
			if (offs_pooled) {                                /* records shorter than 64 bytes of text: this chunk's offsets leave the pool */
				uint64_t *no = (uint64_t *)malloc(cap_off * sizeof(uint64_t));
				memcpy(no, offs, (n + 1) * sizeof(uint64_t));
				offs = no;
				offs_pooled = 0;
			} else {
				offs = (uint64_t *)realloc(offs, cap_off * sizeof(uint64_t));
			}
		}
		offs[n + 1] = offs[n] + (uint64_t)len;
		if (n == 0) first_len = (uint64_t)len;
		else if ((uint64_t)len != first_len) same_len = 0;
		n++;
	}
	c->out.pool_slot = -1;
	c->out.fixed_len = n && same_len ? first_len : 0;
	c->out.text_bytes = (uint64_t)(c->end - c->beg);
	if (c->slot >= 0) {
		if (c->encode && pk.fixed && offs_pooled) {
			c->out.pool_slot = c->slot;                       /* everything stayed in the pool buffer */
		} else {
			/* (a foreign chunk, or one that outgrew its buffer: copy what is pooled out, give the slot back) */
			if (c->encode && pk.fixed) { uint32_t *nw = (uint32_t *)malloc(pk.cap * sizeof(uint32_t)); memcpy(nw, pk.w, pk.cap * sizeof(uint32_t)); pk.w = nw; pk.fixed = 0; }
			if (c->encode && offs_pooled) { uint64_t *no = (uint64_t *)malloc((n + 2) * sizeof(uint64_t)); memcpy(no, offs, (n + 1) * sizeof(uint64_t)); offs = no; }
			sdt_pool_release(c->slot);
			c->slot = -1;
		}
	}
	c->out.words = pk.w;
	c->out.nwords = ((pk.nbases + 15) >> 4) + 4;          /* pk_reserve keeps >= 8 zero words of slack */
	c->out.offsets = offs;
	c->out.nreads = n;
	c->out.owner = c->owner;
	c->out.counted_only = !c->encode;
	c->out.chunk_index = c->index;
	c->out.count_unknown = 0;
	c->out.stream_id = c->out.stream_parity = 0;
	if (!c->encode) {                                      /* a foreign chunk: only the number of records matters */
		free(pk.w);
		free(offs);
		c->out.words = NULL;
		c->out.offsets = NULL;
		c->out.nwords = 0;
	}
}

static void *worker(void *arg)
{
	job_t *J = (job_t *)arg;
	for (;;) {
		/* chunk numbers are taken in order, and so are the pool buffers: the worker of chunk i waits until the buffers of all
		 * earlier chunks have been handed out (J->turn) -- the chunk the consumer is waiting for is never the one left without */
		pthread_mutex_lock(&g_pool.mu);
		const int i = J->next < J->nchunks ? J->next++ : -1;
		int slot = -1;
		if (i >= 0) {
			while (J->turn != i) pthread_cond_wait(&g_pool.cv, &g_pool.mu);
			if (g_pool.on && J->chunks[i].encode)
				slot = pool_acquire_locked(&g_pool.mu, (size_t)(J->chunks[i].end - J->chunks[i].beg));
			J->turn++;
			pthread_cond_broadcast(&g_pool.cv);
		}
		pthread_mutex_unlock(&g_pool.mu);
		if (i < 0) break;
		J->chunks[i].slot = slot;
		parse_chunk(J, &J->chunks[i]);
		pthread_mutex_lock(&J->mu);
		J->chunks[i].done = 1;
		pthread_cond_broadcast(&J->cv);
		pthread_mutex_unlock(&J->mu);
	}
	return NULL;
}

/* first record start at or after p: FASTQ = a line starting with '@' whose line+2 starts with '+';
 * FASTA = a line starting with '>' */
static const char *record_start(const char *base, const char *p, const char *end, int fmt)
{
	if (p <= base) return base;
	/* move to the start of the next line */
	const char *q = line_end(p - 1, end);
	p = q < end ? q + 1 : end;
	while (p < end) {
		if (fmt == 'a') {
			if (*p == '>') return p;
		} else if (*p == '@') {
			const char *l1 = line_end(p, end);
			const char *l2 = l1 < end ? line_end(l1 + 1, end) : end;
			if (l2 < end && l2 + 1 < end && l2[1] == '+') return p;
		}
		const char *e = line_end(p, end);
		p = e < end ? e + 1 : end;
	}
	return end;
}

int sdt_read_file(const char *path, int fmt, int max_read_len, int reverse, int threads, size_t chunk_bytes,
                  sdt_batch_fn fn, void *user, uint64_t *nreads_out)
{
	int fd = open(path, O_RDONLY);
	if (fd < 0) {
		printf("Cannot open %s. Now exit to system...\n", path);
		return -1;
	}
	struct stat st;
	fstat(fd, &st);
	size_t size = (size_t)st.st_size;
	if (nreads_out) *nreads_out = 0;
	if (size == 0) { close(fd); return 0; }
	const char *base = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
	if (base == MAP_FAILED) {
		printf("mmap of %s failed\n", path);
		close(fd);
		return -1;
	}
	madvise((void *)base, size, MADV_SEQUENTIAL);
	const char *end = base + size;
	int nchunks = (int)((size + chunk_bytes - 1) / chunk_bytes);
	job_t J;
	memset(&J, 0, sizeof J);
	J.chunks = (chunk_t *)calloc((size_t)nchunks, sizeof(chunk_t));
	J.fmt = fmt; J.max_read_len = max_read_len; J.reverse = reverse;
	const char *prev = record_start(base, base, end, fmt);
	int nc = 0;
	for (int i = 1; i <= nchunks; i++) {
		const char *cut = i == nchunks ? end : record_start(base, base + (size_t)i * chunk_bytes, end, fmt);
		if (cut > prev) {
			J.chunks[nc].beg = prev;
			J.chunks[nc].end = cut;
			J.chunks[nc].owner = (int)(g_shard_chunk % (uint64_t)g_shard_n);
			J.chunks[nc].encode = g_shard_keep_all || J.chunks[nc].owner == g_shard_rank;
			J.chunks[nc].index = g_shard_chunk;
			sdt_reader_bytes_seen += (uint64_t)(cut - prev);
			if (J.chunks[nc].encode || !g_shard_skip) sdt_reader_bytes_parsed += (uint64_t)(cut - prev);
			g_shard_chunk++;
			nc++;
			prev = cut;
		}
	}
	J.nchunks = nc;
	pthread_mutex_init(&J.mu, NULL);
	pthread_cond_init(&J.cv, NULL);
	if (threads < 1) threads = 1;
	if (threads > nc) threads = nc > 0 ? nc : 1;
	pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
	for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &J);
	int rc = 0;
	uint64_t total = 0;
	for (int i = 0; i < nc; i++) {
		const double t_a = seq_now();
		pthread_mutex_lock(&J.mu);
		while (!J.chunks[i].done) pthread_cond_wait(&J.cv, &J.mu);
		pthread_mutex_unlock(&J.mu);
		const double t_b = seq_now();
		if (rc == 0 && fn(user, &J.chunks[i].out) != 0) rc = -1;
		sdt_reader_wait_ms += t_b - t_a;
		sdt_reader_fn_ms += seq_now() - t_b;
		total += J.chunks[i].out.nreads;
		if (J.chunks[i].out.pool_slot >= 0) {                 /* the callback may have kept the buffer (sdt_pool_take) */
			pthread_mutex_lock(&g_pool.mu);
			const int kept = g_pool.slot[J.chunks[i].out.pool_slot].state == 2;
			pthread_mutex_unlock(&g_pool.mu);
			if (!kept) sdt_pool_release(J.chunks[i].out.pool_slot);
		} else {
			free(J.chunks[i].out.words);
			free(J.chunks[i].out.offsets);
		}
	}
	for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
	free(th);
	free(J.chunks);
	pthread_mutex_destroy(&J.mu);
	pthread_cond_destroy(&J.cv);
	munmap((void *)base, size);
	close(fd);
	if (nreads_out) *nreads_out = total;
	return rc;
}
