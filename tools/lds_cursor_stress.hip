// lds_cursor_stress.hip -- does the LDS cursor protocol of sk_reserve (64-bit ds_add_rtn / ds_wrxchg_rtn on chunk << 32 | pos)
// hand out every slot exactly once when all lanes of a large workgroup hammer ONE cursor?  (the
// level-2 scatter lost records with 1024 lanes per workgroup.)  Build: hipcc --offload-arch=gfx950 -O3 -o lds_cursor_stress
// tools/lds_cursor_stress.hip ; run: ./lds_cursor_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int TPB>
__global__ __launch_bounds__(TPB) void k_stress(unsigned long long *out, int iters, int cap, int delay, unsigned int *g_next)
{
	__shared__ unsigned long long s_cur;
	const int tid = threadIdx.x;
	if (tid == 0)
		s_cur = (0xFFFFFFFFull << 32) | (unsigned)cap;            // "one past the end": the first lane opens a chunk
	__syncthreads();
	for (int i = 0; i < iters; i++) {
		unsigned int chunk = 0, pos = 0;
		bool done = false;
		// (flag form: the allocator's work stays INSIDE the loop body, so a lane spinning on its wave-mate cannot be
		// scheduled ahead of it for ever; the spin is bounded so that a protocol failure shows up as a count, not a hang)
		for (int spin = 0; !done && spin < (1 << 20); spin++) {
			const unsigned long long cur = atomicAdd(&s_cur, 1ULL);
			pos = (unsigned int)cur;
			chunk = (unsigned int)(cur >> 32);
			if (pos < (unsigned)cap) {
				done = true;
			} else if (pos == (unsigned)cap) {
				unsigned int id = atomicAdd(g_next, 1u);               // the "allocation": a global atomic with return
				for (int d = 0; d < delay; d++)
					id += (unsigned int)(__builtin_amdgcn_s_memtime() & 0);     // (keeps the allocator busy for a while)
				atomicExch(&s_cur, ((unsigned long long)id << 32) | 1ULL);
				chunk = id;
				pos = 0;
				done = true;
			}
		}
		out[((size_t)blockIdx.x * TPB + tid) * iters + i] = done ? (((unsigned long long)chunk << 32) | pos) : ~0ULL;
	}
}

template <int TPB> static int run(int blocks, int iters, int cap, int delay)
{
	const size_t n = (size_t)blocks * TPB * iters;
	unsigned long long *d;
	unsigned int *d_next;
	hipMalloc(&d, n * 8);
	hipMalloc(&d_next, 4);
	hipMemset(d_next, 0, 4);
	hipLaunchKernelGGL(k_stress<TPB>, dim3(blocks), dim3(TPB), 0, 0, d, iters, cap, delay, d_next);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	std::vector<unsigned long long> h(n);
	hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
	std::sort(h.begin(), h.end());
	size_t dup = 0, bad = 0;
	for (size_t i = 1; i < n; i++) dup += h[i] == h[i - 1];
	for (size_t i = 0; i < n; i++) bad += (unsigned int)h[i] >= (unsigned)cap || (h[i] >> 32) == 0xFFFFFFFFull;
	size_t stuck = 0;
	for (size_t i = 0; i < n; i++) stuck += h[i] == ~0ULL;
	printf("stuck %zu; ", stuck);
	printf("TPB %4d blocks %3d iters %d cap %d delay %d: %zu slots, %zu handed out twice, %zu invalid\n", TPB, blocks, iters, cap, delay, n, dup, bad);
	fflush(stdout);
	hipFree(d); hipFree(d_next);
	return dup || bad;
}

int main()
{
	int rc = 0;
	for (int delay : {0, 200}) {
		rc |= run<256>(64, 64, 16, delay);
		rc |= run<512>(64, 64, 16, delay);
		rc |= run<768>(64, 64, 16, delay);
		rc |= run<1024>(64, 64, 16, delay);
		rc |= run<1024>(1, 256, 16, delay);
		rc |= run<1024>(256, 64, 32, delay);
	}
	return rc;
}
