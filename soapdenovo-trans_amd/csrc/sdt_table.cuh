// sdt_table.cuh -- device-resident node table for gfx950.
//
// Replaces the reference's KmerSet (inc/newhash.h:65-88, newhash.c) for the counting pass.  What must
// be preserved is the node STATE after all occurrences (survey 9.1):
//     count      = number of occurrences (u32, unsaturated)              newhash.c:75
//     left[b]    = min(63, #occurrences with prev == b)   b = 0..3       newhash.c:77-94, newhash.h:30
//     right[b]   = min(63, #occurrences with next == b)
//     single     = (count == 1)                                          newhash.c:32-40,445
// not the layout (prime-sized table, in-place rehash).  MI355X-first layout:
//
//   Entry<NW> (AoS, 8-byte aligned, 16 / 24 / 40 bytes for 1 / 2 / 4 key words)
//     key[NW]  w[0] most significant.  key[0] doubles as the claim word:
//                 KEY_EMPTY  = ~0      (never a legal high word: at least 2 top bits are unused)
//                 KEY_LOCKED = ~0 - 1  (multi-word keys only: claimed, low words not yet published)
//     val      [63:48] count bits 15..0   [47:24] r_links (4 x 6 bit)   [23:0] l_links (4 x 6 bit)
//   aux[slot]  u32: [15:0] count bits 31..16,  bit 16 linear, bit 17 deleted   (written by scans)
//
// Every occurrence costs ONE 64-bit device-scope atomic on `val`:
//   * both neighbour counters already saturated (or absent)  -> atomicAdd(val, 1<<48); the 16-bit
//     count field sits at the top of the word so its carry falls off the end, and the returned old
//     value tells the one thread that wrapped it to bump aux (count bits 31..16);
//   * otherwise a CAS loop that does the saturating 6-bit increments and the count increment together.
// Counters are monotonic, so a stale observation can only send a thread down the (always correct) CAS
// path; device-scope atomics are resolved at the memory side on gfx950 (per-XCD L2s are not coherent),
// so the atomics, not the plain loads, are authoritative.
#pragma once
#include "sdt_kmer.cuh"
#include "sdt_minimizer.cuh"

namespace sdt {

constexpr uint64_t KEY_EMPTY = ~0ULL;
constexpr uint64_t KEY_LOCKED = ~0ULL - 1ULL;
constexpr uint32_t AUX_LINEAR = 1u << 16;
constexpr uint32_t AUX_DELETED = 1u << 17;
constexpr uint64_t VAL_COUNT_ONE = 1ULL << 48;

template <int NW> struct Entry;
template <> struct alignas(16) Entry<1> {   // 16 B: one global_load_dwordx4 fetches key + val
	uint64_t key[1];
	uint64_t val;
};
template <> struct alignas(32) Entry<2> {   // 32 B, never straddles a 64-B sector: two dwordx4 loads
	uint64_t key[2];
	uint64_t val;
	uint64_t pad;
};
template <> struct alignas(16) Entry<4> {   // 48 B: three dwordx4 loads
	uint64_t key[4];
	uint64_t val;
	uint64_t pad;
};

template <int NW> struct Table {
	Entry<NW> *ent;
	uint32_t *aux;
	uint64_t fslots;      // flat layout: slots (any number since round 5: the home slot is the high word of hash x slots, not hash & mask --
	                      // a table of 2^31 slots for the 0.68 G nodes of the headline workload was a third more to clear and scan than it needs)
	uint64_t *first;      // optional (SDT_FLAG_TRACK_FIRST): smallest ordinal of an occurrence of the key, ~0 = none
	__host__ __device__ uint64_t slots() const { return fslots; }
};

// home slot of a key in the flat table
template <int NW> __device__ inline uint64_t flat_home(const Table<NW> &t, const Key<NW> &key) { return __umul64hi(key_hash<NW>(key), t.fslots); }
__device__ inline uint64_t flat_next(uint64_t slot, uint64_t fslots) { return slot + 1 == fslots ? 0 : slot + 1; }

// Where the probe sequence of `key` starts and the range it wraps in: slot = home, then probe_next() up to `n` times
template <int NW> __device__ inline void probe_begin(const Table<NW> &t, const Key<NW> &key, uint64_t &slot, uint64_t &lo, uint64_t &n)
{
	lo = 0;
	n = t.fslots;
	slot = flat_home<NW>(t, key);
}
__device__ inline uint64_t probe_next(uint64_t slot, uint64_t lo, uint64_t n) { return slot + 1 == lo + n ? lo : slot + 1; }

// read-only look-up: the slot of `key`, or false
template <int NW> __device__ inline bool table_find(const Table<NW> &t, const Key<NW> &key, uint64_t &slot_out)
{
	uint64_t slot, lo, n;
	probe_begin<NW>(t, key, slot, lo, n);
	for (uint64_t probe = 0; probe < n; probe++, slot = probe_next(slot, lo, n)) {
		const Entry<NW> *e = t.ent + slot;
		if (e->key[0] == KEY_EMPTY)
			return false;
		bool same = true;
#pragma unroll
		for (int w = 0; w < NW; w++)
			same = same && e->key[w] == key.w[w];
		if (same) {
			slot_out = slot;
			return true;
		}
	}
	return false;
}

constexpr uint64_t ORD_NONE = ~0ULL;

// first-occurrence tracking: the reference's table layout (and with it the visiting order of the cutting passes
// and the edge ids) is a function of the order in which distinct keys first appear in the read stream (survey
// 7.3-1).  Reads are processed roughly in order, so the stored ordinal is almost always already smaller and the
// atomic is skipped.
__device__ inline void note_first(uint64_t *first, uint64_t slot, uint64_t ord)
{
	if (first && ord != ORD_NONE) {
		if (ord < __hip_atomic_load(first + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			atomicMin((unsigned long long *)(first + slot), (unsigned long long)ord);
	}
}

struct Stats {             // device counters, one cache line each would be nicer; they are cold
	unsigned long long kmers;      // occurrences inserted
	unsigned long long distinct;   // slots claimed
	unsigned long long probe_fail; // inserts that ran out of probes (table too full): fatal
	unsigned long long scratch;    // scan kernels: removed / linear / export cursor
	// locality pipeline (sdt_superkmer.cuh), for sdt_gpu_stage_times
	unsigned long long sk_merges;  // LDS nodes merged into the table (one CAS each)
	unsigned long long sk_spills;  // k-mers that found no LDS slot and took the direct path
	unsigned long long sk_direct;  // k-mers whose record found no chunk and took the direct path
	unsigned long long sk_gens;    // flushes of a full LDS table before its bucket was done
	unsigned long long sk_emitted; // k-mers the level-1 scatter put into records     } conservation: what goes into the pools
	unsigned long long sk_counted; // k-mers k_sk_count took out of level-2 records   } must come out (checked by sync_stats)
	unsigned long long sk_distinct_recs; // records left after k_sk_count's per-tile dedupe (their k-mers are the ones cut, hashed and probed)
	unsigned long long sk_records;  // records that entered the count stage
	unsigned long long sk_distinct_kmers; // k-mers of the distinct records
	unsigned long long sk_cyc1[4]; // k_sk_scatter_reads, thread 0 of every workgroup: clock ticks in tile staging / window minima / run starts / emission
	unsigned long long sk_cyc[4];  // k_sk_count, wave 0 of every workgroup: clock ticks in set-up / tile fill + scan / counting / merging
};

__device__ inline uint64_t ld_relaxed(const uint64_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one occurrence into an existing node
__device__ inline void node_update(uint64_t *val, uint32_t *aux, uint64_t seen, uint32_t prev, uint32_t next)
{
	const int ls = 6 * (int)prev, rs = 24 + 6 * (int)next;
	for (;;) {
		const bool l_done = prev >= 4u || ((seen >> ls) & 63u) >= 63u;
		const bool r_done = next >= 4u || ((seen >> rs) & 63u) >= 63u;
		if (l_done && r_done) {
			const uint64_t old = atomicAdd((unsigned long long *)val, (unsigned long long)VAL_COUNT_ONE);
			if ((old >> 48) == 0xFFFFu)
				atomicAdd(aux, 1u);
			return;
		}
		uint64_t nv = seen + VAL_COUNT_ONE;     // 16-bit count wraps off the top
		if (!l_done) nv += 1ULL << ls;
		if (!r_done) nv += 1ULL << rs;
		const uint64_t got = atomicCAS((unsigned long long *)val, (unsigned long long)seen, (unsigned long long)nv);
		if (got == seen) {
			if ((seen >> 48) == 0xFFFFu)
				atomicAdd(aux, 1u);
			return;
		}
		seen = got;
	}
}

// many occurrences of one key at once: `add` has the layout of `val` (count low 16 | r_links | l_links) with every
// 6-bit field already clamped to 63.  min(63, a + b) per field is exactly what replaying the occurrences one
// by one through update_kmer (newhash.c:71-96) would leave.
__device__ inline void node_merge(uint64_t *val, uint32_t *aux, uint64_t seen, uint64_t add)
{
	for (;;) {
		uint64_t nv = 0;
#pragma unroll
		for (int f = 0; f < 8; f++) {
			const uint32_t a = (uint32_t)(seen >> (6 * f)) & 63u, b = (uint32_t)(add >> (6 * f)) & 63u;
			const uint32_t s = a + b > 63u ? 63u : a + b;
			nv |= (uint64_t)s << (6 * f);
		}
		const uint32_t c = (uint32_t)(seen >> 48) + (uint32_t)(add >> 48);
		nv |= (uint64_t)(c & 0xFFFFu) << 48;
		const uint64_t got = atomicCAS((unsigned long long *)val, (unsigned long long)seen, (unsigned long long)nv);
		if (got == seen) {
			if (c >> 16)
				atomicAdd(aux, c >> 16);
			return;
		}
		seen = got;
	}
}

// Find the slot of `key`, claiming an empty one when the key is new (the probe / claim half of put_kmerset,
// newhash.c:411-462).  On success `slot` is the key's slot and `seen` a (possibly stale) value of its counter
// word: good enough as the first guess of the CAS loops above.  Returns false when the probe budget ran out.
template <int NW>
__device__ inline bool table_locate(const Table<NW> &t, const Key<NW> &key, uint32_t &claimed, uint64_t &slot_out,
                                    uint64_t &seen_out)
{
	uint64_t slot = flat_home<NW>(t, key);
	const uint64_t max_probe = t.fslots < 4096 ? t.fslots : 4096;
	// Multi-word keys: a lane that reads KEY_LOCKED looks again until the claimer -- possibly a lane of its own wave -- has
	// published the low words, so the claimer's stores must stay INSIDE the loop body: its branch ends in `hit = true;
	// continue;` and the loop condition lets it out, not a `return` (an exit path, which the compiler may move behind the
	// loop; tools/lds_cursor_stress.hip hung exactly so with the chunk cursors).  1-word keys have no such wait.
	bool hit = false;
	for (uint64_t probe = 0; probe < max_probe && !hit;) {
		Entry<NW> *e = t.ent + slot;
		uint64_t k0, seen = 0;
		if constexpr (NW == 1) {
			// one 16-byte load: key + val (val may be stale: see node_update)
			const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(e);
			k0 = kv.x;
			seen = kv.y;
		} else {
			// Fast path with plain 16-byte loads.  A published key never changes (write once) and its low words
			// are written before key[0], so: a full match is a match; a different published key[0] is a definite
			// mismatch; anything else (empty, locked, or key[0] equal but low words not (yet) visible in this
			// XCD's L2) goes through the careful agent-scope path below.
			const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(e);
			const ulonglong2 a = q[0];
			k0 = a.x;
			if (k0 != KEY_EMPTY && k0 != KEY_LOCKED) {
				if (k0 != key.w[0]) {
					slot = flat_next(slot, t.fslots);
					probe++;
					continue;
				}
				bool eq = a.y == key.w[1];
				uint64_t v;
				if constexpr (NW == 4) {
					// key[2..3] of a 48-byte entry can sit in another cache line than key[0..1]; a stale copy of
					// that line in this XCD's L2 still holds k_clear's ~0,~0 -- which is a LEGAL low half (64
					// trailing G's, the poly-G tail).  Low words that equal the sentinel are therefore never
					// conclusive here: such candidates take the agent-scope path below.
					const ulonglong2 b = q[1];
					eq = eq && b.x == key.w[2] && b.y == key.w[3] && b.x != KEY_EMPTY && b.y != KEY_EMPTY;
					v = q[2].x;
				} else {
					v = q[1].x;
				}
				if (eq) {
					slot_out = slot;
					seen_out = v;
					return true;
				}
			}
			k0 = ld_relaxed(&e->key[0]);
		}
		if (k0 == KEY_EMPTY) {
			const uint64_t want = NW == 1 ? key.w[0] : KEY_LOCKED;
			const uint64_t old = atomicCAS((unsigned long long *)&e->key[0], (unsigned long long)KEY_EMPTY,
			                               (unsigned long long)want);
			if (old == KEY_EMPTY) {
				claimed++;
				if (NW > 1) {
#pragma unroll
					for (int i = 1; i < NW; i++)
						__hip_atomic_store(&e->key[i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					// Publish key[0] only after the low words have been acknowledged.  All of these are agent-scope
					// (sc1, write-through) stores to ONE entry, and every reader either takes a single-sector
					// snapshot or re-reads with agent-scope loads, so waiting for this lane's own stores is
					// enough; a release fence here would write back the whole XCD L2 (~2-6 us per NEW key,
					// MI355X_MICROARCH.md) and was measured to cap 2-word keys at 3.5 G k-mers/s.
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					__hip_atomic_store(&e->key[0], key.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				slot_out = slot;
				seen_out = 0;
				if constexpr (NW == 1)
					return true;
				hit = true;
				continue;
			}
			k0 = old;        // somebody else got it first: fall through and look at what they put
			seen = 0;
		}
		if (NW > 1 && k0 == KEY_LOCKED)
			continue;        // owner is publishing the low words: look again (no inner spin: SIMT-safe)
		bool same = k0 == key.w[0];
		if (NW > 1 && same) {
			// key[0] was published (release) after the low words, and every load of them below is an
			// agent-scope (sc1, L1-bypassing) load issued after the load of key[0]: no cache
			// invalidate is needed, only that the compiler keeps the order
			__atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
			for (int i = 1; i < NW; i++)
				same = same && (ld_relaxed(&e->key[i]) == key.w[i]);
			if (same)
				seen = ld_relaxed(&e->val);
		}
		if (same) {
			slot_out = slot;
			seen_out = seen;
			return true;
		}
		slot = flat_next(slot, t.fslots);
		probe++;
	}
	return hit;
}

// put_kmerset (newhash.c:411-462) for one record.  Returns false when the probe budget ran out.
template <int NW>
__device__ inline bool table_put(const Table<NW> &t, const Key<NW> &key, uint32_t prev, uint32_t next,
                                 uint32_t &claimed, uint64_t ord = ORD_NONE)
{
	uint64_t slot, seen;
	if (!table_locate<NW>(t, key, claimed, slot, seen))
		return false;
	node_update(&t.ent[slot].val, t.aux + slot, seen, prev, next);
	note_first(t.first, slot, ord);
	return true;
}

// put_kmerset for many occurrences of one key counted elsewhere (LDS): `add` as in node_merge, `hi` = multiples of
// 65536 occurrences on top of add's 16-bit count, `ord` = the smallest ordinal among them
template <int NW>
__device__ inline bool table_merge(const Table<NW> &t, const Key<NW> &key, uint64_t add, uint32_t hi, uint32_t &claimed,
                                   uint64_t ord = ORD_NONE)
{
	uint64_t slot, seen;
	if (!table_locate<NW>(t, key, claimed, slot, seen))
		return false;
	if (add)
		node_merge(&t.ent[slot].val, t.aux + slot, seen, add);
	if (hi)
		atomicAdd(t.aux + slot, hi);
	note_first(t.first, slot, ord);
	return true;
}

// table_merge for a caller that is the ONLY writer of this key's node during the running kernel (k_sk_count on a
// bucket that one workgroup counts alone: every occurrence of a key lives in one bucket).  Only the slot claim of a
// new key races with other writers (other keys, same slot) and stays a CAS; the node itself is read and written
// with agent-scope (cache-coherent, write-through) loads and stores -- no memory-side atomic per merge.
template <int NW>
__device__ inline bool table_merge_owned(const Table<NW> &t, const Key<NW> &key, uint64_t add, uint32_t hi, uint32_t &claimed,
                                         uint64_t ord = ORD_NONE)
{
	uint64_t slot, seen;
	const uint32_t before = claimed;
	if (!table_locate<NW>(t, key, claimed, slot, seen))
		return false;
	uint64_t *val = &t.ent[slot].val;
	if (claimed != before) {
		// the slot was empty a moment ago: k_clear left val = 0, aux = 0, first = none
		__hip_atomic_store(val, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (hi)
			__hip_atomic_store(t.aux + slot, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (t.first && ord != ORD_NONE)
			__hip_atomic_store(t.first + slot, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return true;
	}
	seen = ld_relaxed(val);
	uint64_t nv = 0;
#pragma unroll
	for (int f = 0; f < 8; f++) {
		const uint32_t a = (uint32_t)(seen >> (6 * f)) & 63u, b = (uint32_t)(add >> (6 * f)) & 63u;
		const uint32_t sum = a + b > 63u ? 63u : a + b;
		nv |= (uint64_t)sum << (6 * f);
	}
	const uint32_t cs = (uint32_t)(seen >> 48) + (uint32_t)(add >> 48);
	nv |= (uint64_t)(cs & 0xFFFFu) << 48;
	__hip_atomic_store(val, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const uint32_t up = hi + (cs >> 16);
	if (up) {
		const uint32_t a = __hip_atomic_load(t.aux + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(t.aux + slot, a + up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (t.first && ord != ORD_NONE) {
		if (ord < ld_relaxed(t.first + slot))
			__hip_atomic_store(t.first + slot, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	return true;
}

// ---- owned merges with the loads of several keys in flight ----------------------------------------------------
// table_merge_owned pays three dependent memory round trips per key (probe, re-read of val, store) and k_sk_count's flush
// did them for one slot of the LDS table after the other: ~15 us per flush, a third of the kernel (tick counters, profiles/r3).
// Split in two: ent_load() issues agent-scope loads of everything a merge can need (key words, val, first-occurrence word) --
// the caller issues them for ALL its keys before it looks at any -- and table_merge_owned_at() finishes from the snapshot.
// An owned key has no other writer in this launch, so there is nothing to wait for: a slot that holds another key -- or
// KEY_LOCKED, another workgroup publishing ITS key -- is simply not ours, and the next one is looked at.
// first-occurrence ordinal: a memory-side min that nobody waits for (k_clear leaves ORD_NONE = ~0, so it also serves a new key).
// The load + compare + store it replaces was a second cache line per merge ON the critical path of k_sk_count's flush.
__device__ inline void ord_min_noret(uint64_t *p, uint64_t ord)
{
	(void)__hip_atomic_fetch_min(p, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Publishing a 2-word key into a slot this lane holds (KEY_LOCKED in key[0]): ONE 16-byte store of both words.  An entry of two
// words is 32-byte aligned, so the pair lies in one 16-byte granule of one cache line and reaches L2 as one write: a reader that
// sees the new key[0] sees the new key[1] (it loads key[1] after key[0], in order, from the same line).  The two-step form --
// key[1], wait for the store to be acknowledged, key[0] -- cost a new key a whole memory round trip on the flush's critical path.
static_assert(alignof(Entry<2>) == 32 && sizeof(Entry<2>) == 32, "a 2-word key's pair of words lies in one 16-byte granule of one 32-byte entry");
__device__ inline void store_key_pair(uint64_t *p, uint64_t k0, uint64_t k1)
{
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	u32x4 v;
	v.x = (uint32_t)k0; v.y = (uint32_t)(k0 >> 32); v.z = (uint32_t)k1; v.w = (uint32_t)(k1 >> 32);
	asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

#ifndef SDT_WIDE_ENT_LOAD
#define SDT_WIDE_ENT_LOAD 1
#endif
template <int NW, bool FIRST> struct EntSnap { uint64_t k[NW]; uint64_t v; uint64_t f; bool won; };      // f: the slot's first-occurrence ordinal (FIRST only)

// `claim` (1-word keys): the key is probably new -- its slot claim, a compare-and-swap on the key word, travels TOGETHER with
// the loads instead of after them (the CAS returns the key word either way; on a slot that holds a key it changes nothing).
// A new key then costs one memory round trip, not two; an existing key pays for an atomic it did not need.
template <int NW, bool FIRST> __device__ inline EntSnap<NW, FIRST> ent_load(const Table<NW> &t, uint64_t slot, const Key<NW> &key, bool claim)
{
	EntSnap<NW, FIRST> sn;
	Entry<NW> *e = t.ent + slot;
	sn.won = false;
	if (NW == 1 && claim) {
		sn.k[0] = atomicCAS((unsigned long long *)&e->key[0], (unsigned long long)KEY_EMPTY, (unsigned long long)key.w[0]);
		sn.won = sn.k[0] == KEY_EMPTY;
	} else if (NW == 2 && claim) {
		// 2-word keys: the claim puts KEY_LOCKED there; table_merge_owned_at publishes both key words with one 16-byte store
		sn.k[0] = atomicCAS((unsigned long long *)&e->key[0], (unsigned long long)KEY_EMPTY, (unsigned long long)KEY_LOCKED);
		sn.won = sn.k[0] == KEY_EMPTY;
		sn.k[1] = ld_relaxed(&e->key[NW - 1]);
	} else if (SDT_WIDE_ENT_LOAD) {
		// the whole entry in 16-byte agent-scope loads: every agent-scope load is a request of its own on the way to the memory
		// side (it does not stop in this XCD's L2), also when two of them ask for the same line.  The BYTES fetched did not change
		// (FETCH_SIZE of k_sk_count: 287 GB per step of the 200 M-read workload with two 8-byte loads per entry and with one 16-byte
		// load, profiles/r3 and r4), the time did: 186 -> 174 ms per step -- the flush is bound by requests as much as by bytes
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		if constexpr (NW == 1) {
			u32x4 a;
			asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(e) : "memory");
			sn.k[0] = ((uint64_t)a.y << 32) | a.x;
			sn.v = ((uint64_t)a.w << 32) | a.z;
		} else if constexpr (NW == 2) {
			u32x4 a;
			uint64_t b;
			asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(a), "=&v"(b) : "v"(e) : "memory");
			sn.k[0] = ((uint64_t)a.y << 32) | a.x;
			sn.k[1] = ((uint64_t)a.w << 32) | a.z;
			sn.v = b;
		} else {
			u32x4 a, b;
			uint64_t c;
			asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
			             "global_load_dwordx2 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(e) : "memory");
			sn.k[0] = ((uint64_t)a.y << 32) | a.x;
			sn.k[1] = ((uint64_t)a.w << 32) | a.z;
			sn.k[NW - 2] = ((uint64_t)b.y << 32) | b.x;
			sn.k[NW - 1] = ((uint64_t)b.w << 32) | b.z;
			sn.v = c;
		}
		sn.f = ORD_NONE;
		return sn;
	} else {
#pragma unroll
		for (int i = 0; i < NW; i++)
			sn.k[i] = ld_relaxed(&e->key[i]);
	}
	sn.v = ld_relaxed(&e->val);
	sn.f = ORD_NONE;                                 // (read on demand: see table_merge_owned_at)
	return sn;
}

template <int NW, bool FIRST>
__device__ inline bool table_merge_owned_at(const Table<NW> &t, const Key<NW> &key, uint64_t slot, EntSnap<NW, FIRST> sn, uint64_t add, uint32_t hi,
                                            uint32_t &claimed, uint64_t ord)
{
	const uint64_t max_probe = t.fslots < 4096 ? t.fslots : 4096;
	for (uint64_t probe = 0; probe < max_probe; probe++) {
		Entry<NW> *e = t.ent + slot;
		if (NW <= 2 && sn.won) {
			// the claim that travelled with the loads won the slot: k_clear left val = 0, aux = 0, first = none
			claimed++;
			if (NW == 2) store_key_pair(&e->key[0], key.w[0], key.w[NW - 1]);
			__hip_atomic_store(&e->val, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (hi)
				__hip_atomic_store(t.aux + slot, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (FIRST && ord != ORD_NONE)
				__hip_atomic_store(t.first + slot, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return true;
		}
		bool same = true;
#pragma unroll
		for (int i = 0; i < NW; i++)
			same = same && sn.k[i] == key.w[i];
		if (same) {
			uint64_t nv = 0;
#pragma unroll
			for (int f = 0; f < 8; f++) {
				const uint32_t a = (uint32_t)(sn.v >> (6 * f)) & 63u, b = (uint32_t)(add >> (6 * f)) & 63u;
				const uint32_t sum = a + b > 63u ? 63u : a + b;
				nv |= (uint64_t)sum << (6 * f);
			}
			const uint32_t cs = (uint32_t)(sn.v >> 48) + (uint32_t)(add >> 48);
			nv |= (uint64_t)(cs & 0xFFFFu) << 48;
			__hip_atomic_store(&e->val, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t up = hi + (cs >> 16);
			if (up) {
				const uint32_t a = __hip_atomic_load(t.aux + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(t.aux + slot, a + up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			// the only writer of a key needs no atomic min: load, compare, store -- and the load only for a key that was there
			// already (a third of the merges claim a new slot: they store without ever fetching the ordinal's line)
			if (FIRST && ord != ORD_NONE && ord < ld_relaxed(t.first + slot))
				__hip_atomic_store(t.first + slot, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return true;
		}
		if (sn.k[0] == KEY_EMPTY) {
			const uint64_t old = atomicCAS((unsigned long long *)&e->key[0], (unsigned long long)KEY_EMPTY,
			                               (unsigned long long)(NW == 1 ? key.w[0] : KEY_LOCKED));
			if (old == KEY_EMPTY) {
				claimed++;
				if (NW == 2) {
					store_key_pair(&e->key[0], key.w[0], key.w[NW - 1]);
				} else if (NW > 1) {
#pragma unroll
					for (int i = 1; i < NW; i++)
						__hip_atomic_store(&e->key[i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (as in table_locate: low words before key[0])
					__hip_atomic_store(&e->key[0], key.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				// the slot was empty a moment ago: k_clear left val = 0, aux = 0, first = none
				__hip_atomic_store(&e->val, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (hi)
					__hip_atomic_store(t.aux + slot, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (FIRST && ord != ORD_NONE)
					__hip_atomic_store(t.first + slot, ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return true;
			}
			// somebody else's key (or its KEY_LOCKED) got there first: not ours
		}
		slot = flat_next(slot, t.fslots);
		sn = ent_load<NW, FIRST>(t, slot, key, false);
	}
	return false;
}

} // namespace sdt
