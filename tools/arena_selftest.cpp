// arena_selftest.cpp -- CPU: the bookkeeping of the device memory arena (csrc/sdt_arena.h) under a random load.
// A backing "driver" hands out slabs of address space (no memory behind them); blocks are taken and given back at random, as the
// library does through sdti::dmalloc / dfree, and after every step the book must be airtight:
//   * no two live blocks overlap, every live block lies inside its slab;
//   * live + free ranges of a slab tile it exactly, neighbouring free ranges of one slab are merged;
//   * free_bytes and every slab's `used` agree with the ranges; a trim releases exactly the slabs without live blocks.
// usage: arena_selftest <seed> <steps>      prints "ok ..." or the first violation
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
#include "../soapdenovo-trans_amd/csrc/sdt_arena.h"

using namespace sdt;

static int check(const ArenaBook &A, const char **why)
{
	size_t free_sum = 0;
	std::vector<size_t> used(A.slabs.size(), 0), covered(A.slabs.size(), 0);
	struct Piece { char *a; size_t n; int slab; bool live; };
	std::vector<Piece> all;
	for (auto &kv : A.live) all.push_back(Piece{(char *)kv.first, kv.second.bytes, kv.second.slab, true});
	for (auto &kv : A.free_) { all.push_back(Piece{kv.first, kv.second.bytes, kv.second.slab, false}); free_sum += kv.second.bytes; }
	std::sort(all.begin(), all.end(), [](const Piece &x, const Piece &y) { return x.a < y.a; });
	for (size_t i = 0; i < all.size(); i++) {
		const Piece &p = all[i];
		if (p.slab < 0 || (size_t)p.slab >= A.slabs.size()) { *why = "a range names no slab"; return 1; }
		const ArenaSlab &s = A.slabs[p.slab];
		if (!s.base || p.a < s.base || p.a + p.n > s.base + s.bytes || !p.n) { *why = "a range leaves its slab"; return 1; }
		if (i + 1 < all.size() && p.a + p.n > all[i + 1].a) { *why = "two ranges overlap"; return 1; }
		if (i + 1 < all.size() && !p.live && !all[i + 1].live && all[i + 1].slab == p.slab && p.a + p.n == all[i + 1].a) { *why = "neighbouring free ranges were not merged"; return 1; }
		covered[p.slab] += p.n;
		if (p.live) used[p.slab] += p.n;
	}
	for (size_t i = 0; i < A.slabs.size(); i++) {
		if (!A.slabs[i].base) { if (covered[i]) { *why = "a released slab still has ranges"; return 1; } continue; }
		if (covered[i] != A.slabs[i].bytes) { *why = "live + free do not tile the slab"; return 1; }
		if (used[i] != A.slabs[i].used) { *why = "a slab's used count is off"; return 1; }
	}
	if (free_sum != A.free_bytes) { *why = "free_bytes is off"; return 1; }
	// what mem_info may promise: the unused slabs whole + the largest free range inside a used slab, per device
	for (int dev = 0; dev < 2; dev++) {
		size_t whole = 0, largest = 0;
		for (const Piece &p : all) {
			if (p.live || A.slabs[p.slab].device != dev) continue;
			if (!A.slabs[p.slab].used) whole += p.n;
			else if (p.n > largest) largest = p.n;
		}
		if (A.usable(dev) != whole + largest) { *why = "usable() is off"; return 1; }
		if (A.usable(dev) > A.free_bytes) { *why = "usable() promises more than is free"; return 1; }
	}
	return 0;
}

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
	const int steps = argc > 2 ? atoi(argv[2]) : 20000;
	std::mt19937_64 rng(seed);
	ArenaBook A;
	const size_t GRAIN = 1u << 16;
	char *next_va = (char *)(uintptr_t)0x100000000000ULL;       // the backing allocator: fresh address space, gaps between slabs
	std::vector<std::pair<void *, size_t>> mine;                // what this "library" holds
	size_t slabs_made = 0, reused = 0, trims = 0, released = 0;
	for (int step = 0; step < steps; step++) {
		const unsigned r = (unsigned)(rng() % 100);
		if (r < 55 || mine.empty()) {
			// sizes as the library asks for them: a few MiB to tens of GiB, many repeated
			static const size_t common[] = {1u << 20, 3u << 20, 64u << 20, (size_t)5 << 30, (size_t)8 << 30, (size_t)13 << 30, (size_t)32 << 30};
			size_t want = (rng() % 3) ? common[rng() % 7] : ((rng() % ((size_t)1 << 34)) + (1u << 20));
			want = (want + GRAIN - 1) / GRAIN * GRAIN;
			const int device = (int)(rng() % 2);
			void *p = A.take(want, device);
			if (p) reused++;
			else {
				p = next_va;
				next_va += want + ((rng() % 2) ? 0 : (2u << 20));     // sometimes the next slab is adjacent: ranges must still not merge across
				A.adopt(p, want, device);
				slabs_made++;
			}
			for (auto &m : mine)
				if ((char *)p < (char *)m.first + m.second && (char *)m.first < (char *)p + want) { printf("step %d: a block handed out twice\n", step); return 1; }
			mine.push_back({p, want});
		} else if (r < 95) {
			const size_t i = rng() % mine.size();
			if (!A.give(mine[i].first)) { printf("step %d: the arena does not know a block it handed out\n", step); return 1; }
			mine[i] = mine.back();
			mine.pop_back();
		} else {
			size_t expect = 0;
			for (auto &s : A.slabs) if (s.base && !s.used) expect += s.bytes;
			const size_t got = A.trim([](char *, int) {});
			if (got != expect) { printf("step %d: trim released %zu bytes, %zu were idle\n", step, got, expect); return 1; }
			trims++; released += got;
		}
		if (A.give((void *)(uintptr_t)0x42)) { printf("step %d: a foreign pointer was taken for a block\n", step); return 1; }
		const char *why = nullptr;
		if ((step % 16 == 0 || step + 1 == steps) && check(A, &why)) { printf("step %d: %s\n", step, why); return 1; }
	}
	for (auto &m : mine) A.give(m.first);
	const char *why = nullptr;
	if (check(A, &why)) { printf("at the end: %s\n", why); return 1; }
	size_t idle = 0;
	for (auto &s : A.slabs) if (s.base) idle += s.bytes;
	if (A.trim([](char *, int) {}) != idle || A.free_bytes || !A.free_.empty() || !A.live.empty()) { printf("at the end: the book is not empty after the last trim\n"); return 1; }
	printf("ok: %d steps, %zu slabs from the driver, %zu blocks served from free ranges, %zu trims (%zu GiB released)\n", steps, slabs_made, reused, trims, released >> 30);
	return 0;
}
