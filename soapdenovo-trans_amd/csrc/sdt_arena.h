// sdt_arena.h -- the bookkeeping of the device memory arena (sdt_mem.hip), free of any HIP call so that the CPU suite can drive it
// (tests/test_arena.py through tools/arena_selftest.cpp): slabs obtained from a backing allocator, free ranges by address
// (coalesced inside their slab, never across two), blocks handed out best-fit.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <iterator>
#include <map>
#include <vector>

namespace sdt {

struct ArenaRange { size_t bytes; int slab; };
struct ArenaSlab { char *base; size_t bytes; int device; size_t used; };

struct ArenaBook {
	std::vector<ArenaSlab> slabs;
	std::map<char *, ArenaRange> free_;      // free ranges by address (never spanning two slabs)
	std::map<void *, ArenaRange> live;       // blocks handed out
	size_t free_bytes = 0;

	// best fit among the free ranges of this device; nullptr: nothing fits
	void *take(size_t bytes, int device)
	{
		auto best = free_.end();
		for (auto it = free_.begin(); it != free_.end(); ++it)
			if (it->second.bytes >= bytes && slabs[it->second.slab].device == device && (best == free_.end() || it->second.bytes < best->second.bytes))
				best = it;
		if (best == free_.end()) return nullptr;
		char *p = best->first;
		const ArenaRange r = best->second;
		free_.erase(best);
		if (r.bytes > bytes) free_[p + bytes] = ArenaRange{r.bytes - bytes, r.slab};
		free_bytes -= bytes;
		slabs[r.slab].used += bytes;
		live[p] = ArenaRange{bytes, r.slab};
		return p;
	}
	// a new slab from the backing allocator, handed out whole
	void adopt(void *q, size_t bytes, int device)
	{
		slabs.push_back(ArenaSlab{(char *)q, bytes, device, bytes});
		live[q] = ArenaRange{bytes, (int)slabs.size() - 1};
	}
	// true: p was a block of the arena and is free again
	bool give(void *p)
	{
		auto it = live.find(p);
		if (it == live.end()) return false;
		ArenaRange r = it->second;
		live.erase(it);
		slabs[r.slab].used -= r.bytes;
		free_bytes += r.bytes;
		char *a = (char *)p;
		auto nx = free_.lower_bound(a);
		if (nx != free_.end() && nx->first == a + r.bytes && nx->second.slab == r.slab) {
			r.bytes += nx->second.bytes;
			nx = free_.erase(nx);
		}
		if (nx != free_.begin()) {
			auto pv = std::prev(nx);
			if (pv->first + pv->second.bytes == a && pv->second.slab == r.slab) {
				pv->second.bytes += r.bytes;
				return true;
			}
		}
		free_[a] = r;
		return true;
	}
	// what the arena can really give a new allocation on `device`: the slabs nobody uses (trim returns them to the backing allocator, which
	// may then serve a block of any size) + the largest free range inside a slab that is partly in use (ranges never coalesce across
	// slabs: their SUM says nothing about the largest block that fits)
	size_t usable(int device) const
	{
		size_t whole = 0, largest = 0;
		for (const auto &kv : free_) {
			const ArenaSlab &s = slabs[kv.second.slab];
			if (s.device != device) continue;
			if (!s.used) whole += kv.second.bytes;       // (a slab without live blocks is one free range)
			else if (kv.second.bytes > largest) largest = kv.second.bytes;
		}
		return whole + largest;
	}
	int device_of(void *p) const
	{
		auto it = live.find(p);
		return it == live.end() ? -1 : slabs[it->second.slab].device;
	}
	// slabs nobody uses leave the book; `release(base, device)` gives each back to the backing allocator.  Returns the bytes released
	template <class F> size_t trim(F release)
	{
		size_t out = 0;
		for (size_t i = 0; i < slabs.size(); i++) {
			ArenaSlab &s = slabs[i];
			if (!s.base || s.used) continue;
			free_.erase(s.base);                 // (a slab without live blocks is one free range)
			free_bytes -= s.bytes;
			release(s.base, s.device);
			out += s.bytes;
			s.base = nullptr; s.bytes = 0;
		}
		return out;
	}
};

}  // namespace sdt
