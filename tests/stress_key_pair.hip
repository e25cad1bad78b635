// stress_key_pair.hip -- test program (built and run by tests/test_key_pair_store.py on the GPU box; not part of the product).
// The property the 2-word fast path of csrc/sdt_table.cuh relies on: store_key_pair publishes BOTH words of a 2-word key with one
// 16-byte store into one aligned 16-byte granule, and a reader that loads the granule with one 16-byte agent-scope load (ent_load)
// sees either the claim (KEY_LOCKED in word 0) or the whole key -- never word 0 of the key beside a stale word 1.
// Writers and readers are different workgroups (different CUs and XCDs) on the same table of entries; every round the writers publish
// a fresh pair (k0, k1 = f(k0)) over the lock, the readers hammer the same entries and check the relation on everything that is not
// the lock.  Prints "ok <pairs checked>" or "TORN ...".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "sdt_table.cuh"
using namespace sdt;

__device__ inline uint64_t partner(uint64_t k0) { return (k0 * 0x9E3779B97F4A7C15ULL) ^ 0x5851F42D4C957F2DULL; }

__global__ void k_stress(Entry<2> *ent, uint32_t n, uint32_t rounds, uint32_t writers, unsigned long long *torn, unsigned long long *seen, unsigned int *go)
{
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	const uint32_t wg = blockIdx.x;
	if (wg < writers) {
		for (uint32_t r = 1; r <= rounds; r++) {
			for (uint32_t i = wg * blockDim.x + threadIdx.x; i < n; i += writers * blockDim.x) {
				// claim (as the product does: word 0 becomes the lock), then publish the pair in one store
				__hip_atomic_store(&ent[i].key[0], KEY_LOCKED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint64_t k0 = ((uint64_t)r << 32) | i;
				store_key_pair(&ent[i].key[0], k0, partner(k0));
			}
		}
		if (threadIdx.x == 0) atomicAdd(go, 1u);
	} else {
		unsigned long long bad = 0, ok = 0;
		const uint32_t readers = gridDim.x - writers;
		for (uint32_t it = 0; it < 2000000u && __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < writers; it++) {      // (bounded: a test must end)
			for (uint32_t i = (wg - writers) * blockDim.x + threadIdx.x; i < n; i += readers * blockDim.x) {
				u32x4 a;
				asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(&ent[i]) : "memory");
				const uint64_t k0 = ((uint64_t)a.y << 32) | a.x, k1 = ((uint64_t)a.w << 32) | a.z;
				if (k0 == KEY_LOCKED || k0 == KEY_EMPTY) continue;
				if (k1 != partner(k0)) bad++; else ok++;
			}
		}
		if (bad) atomicAdd(torn, bad);
		if (ok) atomicAdd(seen, ok);
	}
}

int main(int argc, char **argv)
{
	const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u << 16, rounds = argc > 2 ? (uint32_t)atoi(argv[2]) : 200;
	Entry<2> *ent;
	unsigned long long *cnt, h[2] = {0, 0};
	unsigned int *go;
	if (hipMalloc(&ent, (size_t)n * sizeof(Entry<2>)) != hipSuccess || hipMalloc(&cnt, 16) != hipSuccess || hipMalloc(&go, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
	(void)hipMemset(ent, 0xFF, (size_t)n * sizeof(Entry<2>));          // KEY_EMPTY everywhere
	(void)hipMemset(cnt, 0, 16);
	(void)hipMemset(go, 0, 4);
	const uint32_t writers = 64, readers = 192;                 // 256 workgroups: one per CU, all resident together
	hipLaunchKernelGGL(k_stress, dim3(writers + readers), dim3(256), 0, 0, ent, n, rounds, writers, cnt, cnt + 1, go);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
	(void)hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
	if (h[0]) { printf("TORN %llu of %llu pairs\n", h[0], h[0] + h[1]); return 1; }
	printf("ok %llu\n", h[1]);
	return h[1] ? 0 : 3;                                         // (no pair seen at all would prove nothing)
}
