// sdt_pipeline.hpp -- what the host side of the locality pipeline (sdt_pipeline.hip) shares with the multi-GPU exchange
// (sdt_sharded.hip) and the push path (sdt_gpu.hip).  Internal.
#pragma once
#include "sdt_ctx.hpp"

// k-mers per batch at most (pools: ~6 B per k-mer at K = 31).  2^35 since round 6: the 24 G k-mers of the headline workload are ONE batch
// (144 GB of pools beside 36 GB of table on a 288 GB device; sk_alloc halves the batch wherever that does not fit) -- 276.9 -> 264.4 ms per
// step against two batches of 2^34 (profiles/r6): one set of launches, host looks and kernel tails instead of two
static const uint64_t SK_BATCH_MAX_KMERS = 1ULL << clamp_int(sdt_knob_int(sdt_tuning_env("SDT_SK_BATCH_LOG2"), 35), 24, 36);
static const uint32_t SK_ITEM_CHUNKS = 4096;                // level-1 chunks per level-2 work item (4 MiB of records)
static const uint64_t SK_COUNT_KMERS = 1ULL << clamp_int(sdt_knob_int(sdt_tuning_env("SDT_SK_COUNT_KMERS_LOG2"), 29), 20, 36);
static const uint32_t SK_COUNT_PACK_CHUNKS = 64;            // level-2 chunks up to which neighbouring small buckets share a work item (1 K records = two tiles)
// (2048 since round 6 -- one batch per step holds twice the chunks per bucket: with 1024 more buckets were cut into pieces, whose merges are
// compare-and-swaps; count 148.1 -> 145.5 ms at C3, 512: 152.9: profiles/r6/ab_job12*)
static const uint32_t SK_COUNT_ITEM_CHUNKS = (uint32_t)clamp_int(sdt_knob_int(sdt_tuning_env("SDT_SK_COUNT_ITEM_CHUNKS"), 2048), 64, 1 << 24);          // level-2 chunks per k_sk_count work item (16 K records); a bucket within it is counted by ONE workgroup (owned merges)
static const uint32_t SK_MAX_COUNT_LAUNCHES = 4096;          // k-mers per k_sk_count launch (growth bound, see ensure_room)


#define SK_CHK(expr)                                                                                   \
	do {                                                                                               \
		hipError_t e4_ = (expr);                                                                       \
		if (e4_ != hipSuccess)                                                                         \
			return fail(e4_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e4_), __FILE__, __LINE__); \
	} while (0)

struct SkGeo { int mtw, tile_words, hv_words, hv2_words, bits_words; size_t smem; };
SkGeo sk_geo(int K, uint64_t max_read_len);
bool sk_applicable(const sdt_ctx *c, uint64_t max_read_len);
int sk_reset_pool1(sdt_ctx *c);
int sk_alloc(sdt_ctx *c, uint64_t want_kmers, uint64_t per_read);
int sk_list1(sdt_ctx *c);
int sk_split(sdt_ctx *c, const SkPool &src, const uint32_t *list, uint32_t nitems, hipEvent_t after_l2);
int sk_count_all(sdt_ctx *c);
int sk_add_items(sdt_ctx::SkState &k, uint32_t &nitems, uint32_t b, uint32_t lo, uint32_t hi);
int sk_scatter_launch(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nr, uint64_t max_read_len, uint64_t ob,
                      bool allow_direct = true);
uint64_t sk_batch_limit(const sdt_ctx *c);
int sk_scatter(sdt_ctx *c, const uint32_t *d_words, const uint64_t *d_offs, uint64_t nreads, uint64_t max_read_len);
