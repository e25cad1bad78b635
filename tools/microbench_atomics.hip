// microbench_atomics.hip -- what bounds a random-access hash insert on MI355X?
// Measures G ops/s of random 16-B loads, 64-bit atomics (scopes, returning or not), CAS and plain RMW
// on tables of different footprint.  Build: hipcc -O3 --offload-arch=gfx950 -o microbench_atomics microbench_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(ulonglong2 *tbl, uint64_t mask, int iters, uint64_t seed, unsigned long long *sink)
{
	uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	uint64_t acc = 0;
	for (int i = 0; i < iters; i++) {
		uint64_t h = mix(seed + gid * (uint64_t)iters + i);
		if (MODE == 8) h = (h & ~7ULL) | (threadIdx.x & 7);            // 8 lanes share a 128-B group of 8 entries
		uint64_t s = h & mask;
		unsigned long long *v = (unsigned long long *)&tbl[s].y;
		if (MODE == 0) { ulonglong2 e = tbl[s]; acc += e.x + e.y; }                                    // 16-B load
		if (MODE == 1) { __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }    // no-return (result unused)
		if (MODE == 2) { acc += __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // returning
		if (MODE == 3) { unsigned long long exp = 0; __hip_atomic_compare_exchange_strong(v, &exp, 1ULL, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += exp; }
		if (MODE == 4) { ulonglong2 e = tbl[s]; acc += e.x; acc += __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // load + returning add (today's kernel)
		if (MODE == 5) { acc += __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }  // workgroup scope
		if (MODE == 6) { ulonglong2 e = tbl[s]; e.y += 1; tbl[s] = e; }                                // plain RMW (racy)
		if (MODE == 7) { ulonglong2 e = tbl[s]; acc += e.x; __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // load + no-return add
		if (MODE == 8) { acc += __hip_atomic_fetch_add(v, 1ULL << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }       // returning, 8 lanes per 128 B
		if (MODE == 9) { __hip_atomic_fetch_add((unsigned int *)v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // 32-bit no-return
	}
	if (acc == 0x1234567) *sink = acc;
}

template <int MODE> double run(ulonglong2 *tbl, uint64_t slots, const char *name, unsigned long long *sink)
{
	const int iters = 64, blocks = 256 * 8 * 4, threads = 256;
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, tbl, slots - 1, iters, 1ULL, sink);
	CHK(hipDeviceSynchronize());
	CHK(hipEventRecord(a));
	const int reps = 4;
	for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, tbl, slots - 1, iters, 77ULL + r, sink);
	CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
	float ms; CHK(hipEventElapsedTime(&ms, a, b));
	double ops = (double)blocks * threads * iters * reps;
	double g = ops / (ms * 1e-3) / 1e9;
	printf("  %-44s %8.2f G/s\n", name, g);
	return g;
}

int main(int argc, char **argv)
{
	unsigned long long *sink; CHK(hipMalloc(&sink, 8));
	const uint64_t sizes_mb[] = {2, 64, 1024, 16384, 65536};
	for (uint64_t mb : sizes_mb) {
		uint64_t slots = mb * 1024 * 1024 / 16;
		ulonglong2 *tbl; CHK(hipMalloc(&tbl, slots * 16)); CHK(hipMemset(tbl, 0, slots * 16));
		printf("table %llu MiB (%llu slots of 16 B)\n", (unsigned long long)mb, (unsigned long long)slots);
		run<0>(tbl, slots, "random 16-B load", sink);
		run<1>(tbl, slots, "u64 atomic add, no return, agent", sink);
		run<9>(tbl, slots, "u32 atomic add, no return, agent", sink);
		run<2>(tbl, slots, "u64 atomic add, returning, agent", sink);
		run<3>(tbl, slots, "u64 CAS, agent", sink);
		run<4>(tbl, slots, "16-B load + returning add (k_count_reads)", sink);
		run<7>(tbl, slots, "16-B load + no-return add", sink);
		run<5>(tbl, slots, "u64 atomic add, returning, workgroup scope", sink);
		run<6>(tbl, slots, "plain 16-B load + 16-B store (racy)", sink);
		run<8>(tbl, slots, "returning add, 8 lanes per 128-B group", sink);
		CHK(hipFree(tbl));
	}
	return 0;
}
