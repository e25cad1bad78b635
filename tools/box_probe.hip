// box_probe.hip -- four numbers about the box a measurement ran on (hipcc -O3 --offload-arch=gfx950 -o /tmp/box_probe tools/box_probe.hip):
//   1  a returning atomic on ONE address: ns per atomic with one lane issuing them back to back, and with 2048 waves at once
//   2  scattered writes: 208-byte pieces (a group of the level-2 scatter) to 2^18 destinations spread over 16 GiB, GB/s
//   3  a streaming copy of 4 GiB, GB/s
//   4  random 16-byte loads over 16 GiB, G loads/s
// tools/box_modes.sh prints them beside the stage times of bench.py: the level-2 scatter takes 58 ms per step on some boxes and 67 ms on
// others with the same binary (every round since round 2), and this is the search for the property of the box that goes with it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_chain(unsigned long long *ctr, int n, unsigned long long *sink)
{
	unsigned long long s = 0;
	if (threadIdx.x == 0 && blockIdx.x == 0)
		for (int i = 0; i < n; i++) s += atomicAdd(ctr, 1ULL + (s & 1ULL));       // (every atomic waits for the one before)
	if (s == 12345) *sink = s;
}
__global__ void k_many(unsigned long long *ctr, int per_wave, unsigned long long *sink)
{
	unsigned long long s = 0;
	if ((threadIdx.x & 63) == 0)
		for (int i = 0; i < per_wave; i++) s += atomicAdd(ctr, 1ULL);
	if (s == 12345) *sink = s;
}
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every group of 13 lanes writes 13 x 16 = 208 contiguous bytes at the cursor of a pseudo-random destination (2^18 of them, 64 KiB apart)
__global__ void k_scatter(uint4 *dst, uint32_t *cursor, unsigned long long groups)
{
	const unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
	for (unsigned long long g = t / 13; g < groups; g += (unsigned long long)gridDim.x * blockDim.x / 13) {
		const uint32_t b = (uint32_t)(mix(g) & 0x3FFFFu);
		const uint32_t at = (uint32_t)(mix(g * 0x9E3779B97F4A7C15ULL + b) % 300u);            // a slot of the destination's 64 KiB chunk
		dst[(size_t)b * 4096 + (size_t)at * 13 + (t % 13)] = make_uint4((uint32_t)g, b, at, 0);
	}
	(void)cursor;
}
__global__ void k_copy(const uint4 *a, uint4 *b, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_gather(const uint4 *a, size_t mask, unsigned long long per_lane, unsigned long long *sink)
{
	unsigned long long s = 0;
	uint64_t x = mix(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
	for (unsigned long long i = 0; i < per_lane; i++) { const uint4 v = a[x & mask]; s += v.x; x = mix(x + i); }     // (independent addresses: throughput, not latency)
	if (s == 12345) *sink = s;
}

// `box_probe sweep`: the scattered writes alone, on buffers allocated one after the other in ONE process (16 GiB each, some freed in
// between, some kept), and on footprints of 1 / 4 / 16 GiB of one buffer: does the rate go with the allocation?
static int sweep(void)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const size_t big = (size_t)16 << 30;
	const unsigned long long groups = 40000000ULL;
	uint4 *keep[6] = {};
	printf("{\"scatter_sweep_GBps\": [");
	for (int rep = 0; rep < 6; rep++) {
		uint4 *a;
		CHK(hipMalloc(&a, big));
		CHK(hipMemset(a, rep, big));
		CHK(hipDeviceSynchronize());
		float best = 0, ms;
		double r[3];
		for (int t = 0; t < 3; t++) {
			CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(256), 0, 0, a, (uint32_t *)nullptr, groups); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			CHK(hipEventElapsedTime(&ms, e0, e1));
			r[t] = (double)groups * 208 / (ms * 1e6);
			if (r[t] > best) best = (float)r[t];
		}
		printf("%s[%.0f, %.0f, %.0f]", rep ? ", " : "", r[0], r[1], r[2]);
		if (rep & 1) keep[rep] = a; else CHK(hipFree(a));           // (every second buffer stays: the next one gets other pages)
	}
	printf("]}\n");
	for (int i = 0; i < 6; i++) if (keep[i]) CHK(hipFree(keep[i]));
	return 0;
}

// `box_probe vmm <granule MiB> <shuffle 0|1>`: the same scattered writes into 16 GiB of virtual addresses backed by physical granules
// of the given size (hipMemCreate), mapped in the order they were created or in a shuffled order
static int vmm(size_t gran_mib, int shuffle)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const size_t big = (size_t)16 << 30;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t min_gran = 0;
	CHK(hipMemGetAllocationGranularity(&min_gran, &prop, hipMemAllocationGranularityMinimum));
	size_t gran = gran_mib << 20;
	if (gran < min_gran) gran = min_gran;
	gran = (gran + min_gran - 1) / min_gran * min_gran;
	const size_t n = big / gran;
	hipMemGenericAllocationHandle_t *h = (hipMemGenericAllocationHandle_t *)malloc(n * sizeof *h);
	size_t *order = (size_t *)malloc(n * sizeof(size_t));
	for (size_t i = 0; i < n; i++) { CHK(hipMemCreate(&h[i], gran, &prop, 0)); order[i] = i; }
	if (shuffle) { uint64_t x = 88172645463325252ULL; for (size_t i = n - 1; i > 0; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const size_t j = x % (i + 1); const size_t t = order[i]; order[i] = order[j]; order[j] = t; } }
	void *va = nullptr;
	CHK(hipMemAddressReserve(&va, big, 0, nullptr, 0));
	for (size_t i = 0; i < n; i++) CHK(hipMemMap((char *)va + i * gran, gran, 0, h[order[i]], 0));
	hipMemAccessDesc acc = {};
	acc.location.type = hipMemLocationTypeDevice;
	acc.location.id = 0;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	CHK(hipMemSetAccess(va, big, &acc, 1));
	CHK(hipMemset(va, 3, big));
	CHK(hipDeviceSynchronize());
	const unsigned long long groups = 40000000ULL;
	double r[3];
	float ms;
	for (int t = 0; t < 3; t++) {
		CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(256), 0, 0, (uint4 *)va, (uint32_t *)nullptr, groups); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		CHK(hipEventElapsedTime(&ms, e0, e1));
		r[t] = (double)groups * 208 / (ms * 1e6);
	}
	printf("{\"vmm_granule_MiB\": %zu, \"granules\": %zu, \"shuffled\": %d, \"scatter_GBps\": [%.0f, %.0f, %.0f]}\n", gran >> 20, n, shuffle, r[0], r[1], r[2]);
	CHK(hipMemUnmap(va, big));
	for (size_t i = 0; i < n; i++) CHK(hipMemRelease(h[i]));
	CHK(hipMemAddressFree(va, big));
	return 0;
}

// `box_probe pieces`: the scattered writes with pieces of 96 ... 1536 bytes (L lanes x 16 bytes each, to 2^18 destinations over 16 GiB)
template <int L>
__global__ void k_scatter_l(uint4 *dst, unsigned long long groups)
{
	const unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x / L;
	if (t / L >= stride) return;
	for (unsigned long long g = t / L; g < groups; g += stride) {
		const uint32_t b = (uint32_t)(mix(g) & 0x3FFFFu);
		const uint32_t at = (uint32_t)(mix(g * 0x9E3779B97F4A7C15ULL + b) % (4096u / L));
		dst[(size_t)b * 4096 + (size_t)at * L + (t % L)] = make_uint4((uint32_t)g, b, at, 0);
	}
}
template <int L> static int pieces_one(uint4 *a, hipEvent_t e0, hipEvent_t e1)
{
	const unsigned long long bytes = 8ULL << 30, groups = bytes / (16ULL * L);
	float ms, best = 1e9f;
	for (int t = 0; t < 3; t++) {
		CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter_l<L>, dim3(2048), dim3(256), 0, 0, a, groups); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		CHK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	printf("%s\"%d\": %.0f", L == 1 ? "" : ", ", L * 16, (double)groups * 16 * L / (best * 1e6));
	return 0;
}
static int pieces(void)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	uint4 *a;
	CHK(hipMalloc(&a, (size_t)16 << 30));
	CHK(hipMemset(a, 1, (size_t)16 << 30));
	CHK(hipDeviceSynchronize());
	printf("{\"scatter_GBps_by_piece_bytes\": {");
	if (pieces_one<1>(a, e0, e1) || pieces_one<2>(a, e0, e1) || pieces_one<3>(a, e0, e1) || pieces_one<4>(a, e0, e1) || pieces_one<6>(a, e0, e1) || pieces_one<8>(a, e0, e1) || pieces_one<12>(a, e0, e1) || pieces_one<16>(a, e0, e1) || pieces_one<24>(a, e0, e1) || pieces_one<32>(a, e0, e1) || pieces_one<64>(a, e0, e1) || pieces_one<96>(a, e0, e1)) return 1;
	printf("}}\n");
	return 0;
}

int main(int argc, char **argv)
{
	if (argc > 1 && argv[1][0] == 'p') return pieces();
	if (argc > 1 && argv[1][0] == 's') return sweep();
	if (argc > 3 && argv[1][0] == 'v') return vmm((size_t)atoi(argv[2]), atoi(argv[3]));
	unsigned long long *ctr, *sink;
	CHK(hipMalloc(&ctr, 8)); CHK(hipMalloc(&sink, 8));
	CHK(hipMemset(ctr, 0, 8));
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	float ms;
	hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, ctr, 1000, sink);
	CHK(hipDeviceSynchronize());
	const int N1 = 200000;
	CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, ctr, N1, sink); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double chain_ns = ms * 1e6 / N1;
	const int PW = 512, BL = 512;                                   // 512 blocks x 4 waves x 512 atomics = 1 M
	CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_many, dim3(BL), dim3(256), 0, 0, ctr, PW, sink); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double many_ns = ms * 1e6 / ((double)BL * 4 * PW);
	const size_t big = (size_t)16 << 30;
	uint4 *a, *b;
	CHK(hipMalloc(&a, big)); CHK(hipMalloc(&b, (size_t)4 << 30));
	CHK(hipMemset(a, 1, big)); CHK(hipMemset(b, 2, (size_t)4 << 30));
	CHK(hipDeviceSynchronize());
	const unsigned long long groups = 40000000ULL;                  // 8.3 GB of 208-byte pieces
	hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(256), 0, 0, a, (uint32_t *)nullptr, groups / 8);
	CHK(hipDeviceSynchronize());
	CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(256), 0, 0, a, (uint32_t *)nullptr, groups); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double scatter_gbs = (double)groups * 208 / (ms * 1e6);
	CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, ((size_t)4 << 30) / 16); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double copy_gbs = 2.0 * 4.294967296 / (ms * 1e-3);
	const unsigned long long per_lane = 256;
	CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, 0, a, big / 16 - 1, per_lane, sink); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	CHK(hipEventElapsedTime(&ms, e0, e1));
	const double gather_g = 2048.0 * 256 * per_lane / (ms * 1e6);
	printf("{\"atomic_chain_ns\": %.2f, \"atomic_one_address_ns\": %.2f, \"scatter_208B_GBps\": %.1f, \"copy_GBps\": %.1f, \"gather_16B_Gps\": %.2f}\n",
	       chain_ns, many_ns, scatter_gbs, copy_gbs, gather_g);
	return 0;
}
