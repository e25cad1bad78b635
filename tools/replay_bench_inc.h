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
		if (i + RP_AHEAD < m) RP_PF(&t[keys[i + RP_AHEAD] % size]);    /* its home, unless the table grows first */
		uint64_t h = keys[i] % size;
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

