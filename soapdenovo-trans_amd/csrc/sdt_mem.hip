// sdt_mem.hip -- device memory of libsdt_gpu.so: every hipMalloc / hipFree of the library ends here (sdt_internal.hpp).
//
// Why an arena.  What a hipMalloc costs on the box depends on the state its memory is in: a block that was never used comes in
// 0.3 ms, 32 GiB of it; a block that some process -- this one included -- gave back a moment ago has to be cleared by the
// driver first, at ~33 GiB/s on a good day and behind the clearing of everything else that was freed (measured: 1-4.7 s for the
// 48 GiB of the locality pipeline's pools, 3.3 s for the first 5 GiB the layout asked for right after those pools were
// released; profiles/r4/README.md).  A pregraph run allocates ~250 GiB in all but never more than ~120 GiB at a time, so blocks
// of at least 1 MiB are kept when they are freed and handed out again: the driver sees the peak, once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <map>
#include <mutex>
#include <vector>
#include "sdt_internal.hpp"
#include "sdt_arena.h"
#undef hipMalloc
#undef hipFree
#undef hipHostMalloc

namespace {

bool mem_timing() { static const bool t = sdt_env("SDT_TIMING") != nullptr; return t; }
double mem_now() { struct timespec a; clock_gettime(CLOCK_MONOTONIC, &a); return a.tv_sec * 1e3 + a.tv_nsec * 1e-6; }
void mem_report(const char *what, double gib, double ms, const char *file, int line)
{
	if (!mem_timing() || ms <= 5.0) return;
	const char *f = strrchr(file, '/');
	fprintf(stderr, "[device] %s of %.2f GiB took %.1f ms (%s:%d)\n", what, gib, ms, f ? f + 1 : file, line);
}

constexpr size_t ARENA_MIN = 1u << 20;       // smaller blocks go straight to the runtime (its own sub-allocator serves them)
constexpr size_t ARENA_GRAIN = 1u << 16;

struct Arena : sdt::ArenaBook {
	std::mutex mu;
	bool off() { static const bool o = sdt_env("SDT_NO_ARENA") != nullptr; return o; }
	// slabs nobody uses go back to the driver (all devices); returns the bytes released
	size_t trim_to_driver()
	{
		int cur = 0;
		(void)hipGetDevice(&cur);
		const size_t out = trim([](char *base, int device) {
			(void)hipSetDevice(device);
			(void)hipFree(base);
		});
		(void)hipSetDevice(cur);
		return out;
	}
};

Arena &arena() { static Arena *a = new Arena; return *a; }      // (never destroyed: contexts may outlive static destructors)

}  // namespace

hipError_t sdti::dmalloc(void **p, size_t bytes, const char *file, int line)
{
	Arena &A = arena();
	if (bytes < ARENA_MIN || A.off()) {
		const double t0 = mem_now();
		const hipError_t e = hipMalloc(p, bytes);
		mem_report("hipMalloc", (double)bytes / (1 << 30), mem_now() - t0, file, line);
		return e;
	}
	int device = 0;
	hipError_t e = hipGetDevice(&device);
	if (e != hipSuccess) return e;
	const size_t want = (bytes + ARENA_GRAIN - 1) / ARENA_GRAIN * ARENA_GRAIN;
	std::lock_guard<std::mutex> lock(A.mu);
	if ((*p = A.take(want, device)) != nullptr) return hipSuccess;
	const double t0 = mem_now();
	void *q = nullptr;
	e = hipMalloc(&q, want);
	if (e == hipErrorOutOfMemory && A.trim_to_driver()) {
		(void)hipGetLastError();
		e = hipMalloc(&q, want);
	}
	mem_report("hipMalloc", (double)want / (1 << 30), mem_now() - t0, file, line);
	if (e != hipSuccess) { *p = nullptr; return e; }
	A.adopt(q, want, device);
	*p = q;
	return hipSuccess;
}

hipError_t sdti::dfree(void *p, const char *file, int line)
{
	if (!p) return hipSuccess;
	Arena &A = arena();
	{
		std::lock_guard<std::mutex> lock(A.mu);
		if (A.live.count(p)) {
			// hipFree waits for the device before it lets go of a block; the callers rely on that (kernels of another stream may
			// still read what is freed here).  The device that OWNS the block: a process may hold contexts on several.
			int cur = 0;
			(void)hipGetDevice(&cur);
			const int dev = A.device_of(p);
			if (dev >= 0 && dev != cur) (void)hipSetDevice(dev);
			const hipError_t e = hipDeviceSynchronize();
			if (dev >= 0 && dev != cur) (void)hipSetDevice(cur);
			(void)A.give(p);
			return e;
		}
	}
	const double t0 = mem_now();
	const hipError_t e = hipFree(p);
	mem_report("hipFree", 0.0, mem_now() - t0, file, line);
	return e;
}

hipError_t sdti::hmalloc(void **p, size_t bytes, unsigned flags, const char *file, int line)
{
	const double t0 = mem_now();
	const hipError_t e = hipHostMalloc(p, bytes, flags);
	mem_report("hipHostMalloc", (double)bytes / (1 << 30), mem_now() - t0, file, line);
	return e;
}

hipError_t sdti::mem_info(size_t *free_b, size_t *total_b)
{
	const hipError_t e = hipMemGetInfo(free_b, total_b);
	if (e != hipSuccess) return e;
	// + what the arena can give a new block on THIS device: its unused slabs and the largest range inside a used one (not the sum
	// of all free ranges of all devices: a pool sized to 60 % of that could ask for a block no range holds, with the driver's memory
	// pinned by small long-lived blocks carved out of the large slabs)
	int device = 0;
	(void)hipGetDevice(&device);
	Arena &A = arena();
	std::lock_guard<std::mutex> lock(A.mu);
	*free_b += A.usable(device);
	return hipSuccess;
}

size_t sdti::mem_trim(void)
{
	Arena &A = arena();
	std::lock_guard<std::mutex> lock(A.mu);
	return A.trim_to_driver();
}
