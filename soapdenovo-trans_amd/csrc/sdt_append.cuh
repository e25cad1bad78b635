// sdt_append.cuh -- records appended by many lanes to one array WITHOUT one atomic per wave on one address.
//
// A returning atomic on ONE address costs about 15 ns on this part however it is issued (measured: 23.6 M wave-level reservations of
// k_layout_keys = 376 ms, 6.5 M of the replay's evaluation = 97 ms): a kernel that scans the node table and appends a record for 1-2 %
// of the slots makes one such reservation per wave and iteration and spends most of its time on them.  Here a WAVE takes a chunk of
// AP_CH record slots at a time from a chunk cursor (one atomic per chunk) and hands the slots out from a counter of its own in LDS.
// The lanes of a wave that reach ap_append together (whatever the divergence around it) are served by one of them; lanes of the same
// wave on another path run at another time, lanes of other waves have another counter: the counter needs no atomics.
// The array has holes (the unused tail of every wave's last chunk): `fill[chunk]` says how many slots of a chunk are used and
// k_ap_compact packs the records (exclusive scan of the fills, chunk order kept); a list of single words can instead mark the holes
// (ap_finish_mark) and have its readers skip them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdt {

constexpr uint32_t AP_CH = 256;
constexpr unsigned long long AP_NONE = ~0ULL;

struct WaveApp {                       // one per wave of the workgroup, in LDS
	unsigned long long chunk;
	uint32_t used, pad;
};

struct ApOut {
	unsigned long long *cursor;        // chunks taken so far (keeps counting past cap_chunks: the caller retries with that many)
	unsigned long long cap_chunks;
	uint32_t *fill;                    // per chunk: slots in use (records that are packed afterwards), or nullptr
	unsigned long long *mark;          // ... or the list itself: its unused slots are set to AP_NONE (single words, readers skip them)
};

__device__ inline void ap_init(WaveApp *app)          // every lane, first thing in the kernel (app: TPB / 64 entries in LDS)
{
	if ((threadIdx.x & 63u) == 0) {
		volatile WaveApp *a = app + (threadIdx.x >> 6);
		a->chunk = AP_NONE;
		a->used = AP_CH;
	}
}

// the slot of this lane's record (AP_NONE: past the capacity).  Any subset of a wave's lanes may call it together.
__device__ inline unsigned long long ap_append(WaveApp *app, const ApOut &o)
{
	volatile WaveApp *a = app + (threadIdx.x >> 6);
	const unsigned long long here = __ballot(1);
	const int lane = (int)__lane_id(), leader = __ffsll((long long)here) - 1;
	const uint32_t n = (uint32_t)__popcll(here);
	unsigned long long base = 0;
	if (lane == leader) {
		uint32_t used = a->used;
		unsigned long long chunk = a->chunk;
		if (used + n > AP_CH) {
			if (chunk < o.cap_chunks) {                    // (a chunk is closed with up to 63 slots to spare)
				if (o.fill) o.fill[chunk] = used;
				if (o.mark) for (uint32_t j = used; j < AP_CH; j++) o.mark[chunk * AP_CH + j] = AP_NONE;
			}
			chunk = atomicAdd(o.cursor, 1ULL);
			used = 0;
			a->chunk = chunk;
		}
		a->used = used + n;
		base = chunk < o.cap_chunks ? chunk * AP_CH + used : AP_NONE;
	}
	base = __shfl(base, leader);
	return base == AP_NONE ? AP_NONE : base + (unsigned long long)__popcll(here & ((1ULL << lane) - 1ULL));
}

// last thing in the kernel, every lane: the fill of the wave's open chunk
__device__ inline void ap_finish(WaveApp *app, const ApOut &o)
{
	volatile WaveApp *a = app + (threadIdx.x >> 6);
	if ((threadIdx.x & 63u) == 0 && o.fill && a->chunk < o.cap_chunks) o.fill[a->chunk] = a->used;
}

// ... or, for a list of single words: the unused slots of the wave's open chunk are marked AP_NONE
__device__ inline void ap_finish_mark(WaveApp *app, const ApOut &o)
{
	volatile WaveApp *a = app + (threadIdx.x >> 6);
	const unsigned long long chunk = a->chunk;
	if (chunk >= o.cap_chunks || !o.mark) return;
	for (uint32_t j = a->used + (threadIdx.x & 63u); j < AP_CH; j += 64u) o.mark[chunk * AP_CH + j] = AP_NONE;
}

// Dense output when EVERY thread of the workgroup takes part (a scan loop whose bounds depend on the workgroup only): each thread says
// how many slots it wants for its next few items, the workgroup makes ONE reservation.  Returns the thread's first slot.
// scratch: 1 + blockDim.x / 64 words of LDS.
__device__ inline unsigned long long ap_block_reserve(uint32_t mine, unsigned long long *cursor, unsigned long long *scratch)
{
	const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
	uint32_t inc = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(inc, d);
		if (lane >= d) inc += o;
	}
	__syncthreads();                                   // (the scratch of the call before has been read)
	if (lane == 63) scratch[1 + wave] = inc;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long tot = 0;
		for (unsigned w = 0; w < (blockDim.x >> 6); w++) tot += scratch[1 + w];
		scratch[0] = tot ? atomicAdd(cursor, tot) : 0ULL;
	}
	__syncthreads();
	unsigned long long at = scratch[0] + (inc - mine);
	for (int w = 0; w < wave; w++) at += scratch[1 + w];
	return at;
}

// records of `stride` words: chunk c's first fill[c] slots go to out[off[c] ...] (off = exclusive scan of fill)
static __global__ __launch_bounds__(256) void k_ap_compact(const uint64_t *__restrict__ chunks, const uint32_t *__restrict__ fill, const uint32_t *__restrict__ off,
                                                    unsigned long long n_chunks, int stride, uint64_t *__restrict__ out)
{
	const unsigned long long total = n_chunks * AP_CH;
	for (unsigned long long g = blockIdx.x * 256ULL + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * 256ULL) {
		const unsigned long long c = g / AP_CH;
		const uint32_t j = (uint32_t)(g % AP_CH);
		if (j >= fill[c]) continue;
		const uint64_t *src = chunks + g * (unsigned long long)stride;
		uint64_t *dst = out + ((unsigned long long)off[c] + j) * (unsigned long long)stride;
		for (int w = 0; w < stride; w++) dst[w] = src[w];
	}
}

}  // namespace sdt
