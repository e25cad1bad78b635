// sdt_count_plan.h -- the work items and launches of the count stage as a PURE host function of the level-2 chunk lists.
//
// After the level-2 scatter the host knows, per final bucket f, where its chunks lie in the chunk list (off2[f] .. off2[f + 1])
// and how many k-mers they hold (kpre2[f + 1] - kpre2[f]).  k_sk_count's workgroups take work items first come first served;
// an item is a run [c0, c1) of the list:
//   * a bucket of at most `item_chunks` chunks is one item, flagged WHOLE (top bit of c1): no other item of the launch holds
//     keys of its buckets: the workgroup that takes it merges into the node table WITHOUT atomics (plain read-modify-write, a
//     compare-and-swap only for the claim of a new key's slot);
//   * a larger bucket is cut into pieces of `item_chunks` chunks, not flagged;
//   * buckets of at most `pack_chunks` chunks share an item with their neighbours (their chunks are adjacent in the list, empty
//     buckets in between do not matter) as long as the item stays within `pack_chunks` chunks, within ONE level-1 bucket and
//     within a span of SK_PLAN_MAX_SPAN final buckets per-item costs are paid once;
//   * an item is FOUR words: c0, c1 | WHOLE, first final bucket, last final bucket;
//   * within a launch the items are handed out LARGEST FIRST (by power-of-two size class, list order within a class): a launch
//     ends when its slowest workgroup does, and an item of 1024 chunks taken last kept 500 workgroups waiting for ~0.7 ms --
//     45 launches per step of the 200 M-read workload, 30 ms of its 170 ms (profiles/r4);
//   * launches are cut between buckets: a launch holds at most `limit` k-mers (`first_limit` for the first one, whose rate of
//     new nodes sizes the others) unless a single bucket is larger, and an item never spans two launches.
// sk_count_all (sdt_gpu.hip) calls this; tests/test_count_plan.py checks it on the CPU through sdt_sk_plan_count_items.
#pragma once
#include <stdint.h>
#include <vector>

namespace sdt {

constexpr uint32_t SK_ITEM_WHOLE = 0x80000000u;
constexpr uint32_t SK_PLAN_MAX_SPAN = 64;        // final buckets an item may span 
constexpr uint32_t SK_PLAN_L2_BUCKETS = 1024;    // final buckets per level-1 bucket (= SK_NB2)
constexpr int SK_ITEM_WORDS = 4;

// items [i0, i1) largest first: a counting sort by size class (position of the highest bit of the chunk count), stable within a class
inline void sk_plan_largest_first(uint32_t *items, uint32_t i0, uint32_t i1, std::vector<uint32_t> &tmp)
{
	if (i1 - i0 < 2)
		return;
	constexpr int W = SK_ITEM_WORDS;
	uint32_t cnt[33] = {0};
	auto cls = [&](uint32_t i) { const uint32_t n = (items[W * i + 1] & ~SK_ITEM_WHOLE) - items[W * i]; return n ? 32u - (uint32_t)__builtin_clz(n) : 0u; };
	for (uint32_t i = i0; i < i1; i++)
		cnt[cls(i)]++;
	uint32_t start[33], acc = 0;
	for (int c = 32; c >= 0; c--) { start[c] = acc; acc += cnt[c]; }
	tmp.resize((size_t)W * (i1 - i0));
	for (uint32_t i = i0; i < i1; i++) {
		const uint32_t at = start[cls(i)]++;
		for (int w = 0; w < W; w++)
			tmp[W * at + w] = items[W * i + w];
	}
	for (uint32_t j = 0; j < W * (i1 - i0); j++)
		items[W * i0 + j] = tmp[j];
}

// items: SK_ITEM_WORDS words per item (c0, c1 | SK_ITEM_WHOLE, first bucket, last bucket); first_item[l] = first item of launch l (first_item[*nlaunches] = *nitems);
// launch_kmers[l] = its k-mers.  Returns false when an output array is too small.
inline bool sk_plan_count_items(const uint32_t *off2, const uint64_t *kpre2, uint32_t nbuckets, uint64_t first_limit, uint64_t limit,
                                uint32_t max_launches, uint32_t pack_chunks, uint32_t item_chunks, uint32_t *items, uint32_t items_cap,
                                uint32_t *first_item, uint64_t *launch_kmers, uint32_t launches_cap, uint32_t *nitems, uint32_t *nlaunches)
{
	constexpr int W = SK_ITEM_WORDS;
	uint32_t nci = 0, nl = 0;
	uint64_t acc = 0;
	std::vector<uint32_t> tmp;
	bool pack_open = false;
	uint32_t pack_c0 = 0, pack_f0 = 0;
	if (launches_cap < 1)
		return false;
	first_item[0] = 0;
	for (uint32_t f = 0; f < nbuckets; f++) {
		const uint64_t km = kpre2[f + 1] - kpre2[f];
		const uint64_t lim = nl == 0 ? first_limit : limit;
		if (acc && acc + km > lim && nl + 1 < max_launches) {
			if (nl + 2 > launches_cap)
				return false;
			launch_kmers[nl++] = acc;
			first_item[nl] = nci;
			sk_plan_largest_first(items, first_item[nl - 1], nci, tmp);
			acc = 0;
			pack_open = false;                           // (an item belongs to one launch)
		}
		acc += km;
		const uint32_t nch = off2[f + 1] - off2[f];
		if (nch <= pack_chunks && pack_open && off2[f + 1] - pack_c0 <= pack_chunks && f - pack_f0 < SK_PLAN_MAX_SPAN &&
		    f / SK_PLAN_L2_BUCKETS == pack_f0 / SK_PLAN_L2_BUCKETS) {
			if (nch) {
				items[W * (nci - 1) + 1] = off2[f + 1] | SK_ITEM_WHOLE;
				items[W * (nci - 1) + 3] = f;
			}
			continue;
		}
		pack_open = false;
		if (!nch)
			continue;
		const uint32_t whole = nch <= item_chunks ? SK_ITEM_WHOLE : 0u;
		for (uint32_t c0 = off2[f]; c0 < off2[f + 1]; c0 += item_chunks) {
			const uint32_t c1 = c0 + item_chunks < off2[f + 1] ? c0 + item_chunks : off2[f + 1];
			if (nci >= items_cap)
				return false;
			items[W * nci] = c0;
			items[W * nci + 1] = c1 | whole;
			items[W * nci + 2] = f;
			items[W * nci + 3] = f;
			nci++;
		}
		if (nch <= pack_chunks) {                        // the next small buckets may join this item
			pack_open = true;
			pack_c0 = off2[f];
			pack_f0 = f;
		}
	}
	if (nl + 2 > launches_cap)
		return false;
	launch_kmers[nl++] = acc;
	first_item[nl] = nci;
	sk_plan_largest_first(items, first_item[nl - 1], nci, tmp);
	*nitems = nci;
	*nlaunches = nl;
	return true;
}

} // namespace sdt
