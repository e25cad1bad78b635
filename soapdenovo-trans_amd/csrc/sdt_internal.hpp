// sdt_internal.hpp -- what the translation units of libsdt_gpu.so share besides the public ABI (include/sdt_gpu.h).
//   sdt_gpu.hip        pass 1 (direct kernel family, locality pipeline), table scans, second read pass, multi-GPU, map stage
//   sdt_mem.hip        device memory: the arena behind every hipMalloc / hipFree of the library
//   sdt_gpu_graph.hip  graph phases on the device mirror: layout (visiting order), dry runs of the cutting passes with
//                      the components of their commits, port walks of kmer2edges
// The context itself (struct sdt_ctx) stays private to sdt_gpu.hip; the graph unit sees it through GraphView.
#pragma once
#include "sdt_knobs.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/sdt_gpu.h"
#include "sdt_kmer.cuh"
#include "sdt_table.cuh"

constexpr int TPB = 256;           // 4 waves: block size of every scan-style kernel

namespace sdti {

// error plumbing: sets the message sdt_gpu_last_error() returns (thread local), returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIPCHK(expr)                                                                                         \
	do {                                                                                                     \
		hipError_t e_ = (expr);                                                                              \
		if (e_ != hipSuccess)                                                                                \
			return sdti::fail(e_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s failed: %s (%s:%d)",    \
			                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                             \
	} while (0)

// every device allocation of the library goes through here (sdt_mem.hip): blocks of 1 MiB and more are kept in an arena when they
// are freed and handed out again; under SDT_TIMING the calls into the runtime that take more than 5 ms are reported
hipError_t dmalloc(void **p, size_t bytes, const char *file, int line);
hipError_t dfree(void *p, const char *file, int line);
hipError_t hmalloc(void **p, size_t bytes, unsigned flags, const char *file, int line);
hipError_t mem_info(size_t *free_b, size_t *total_b);      // hipMemGetInfo + what the arena holds
size_t mem_trim(void);                                     // slabs of the arena that nobody uses go back to the driver

struct GraphExt;                   // state of the graph unit, owned by the context (sdt_gpu_graph.hip)
void graph_ext_free(GraphExt *gx);
// the path words sdt_gpu_build_edges left on the device (n words, one per node index), or nullptr; the caller frees them
uint64_t *graph_take_path_words(GraphExt *gx, uint64_t n);

struct GraphView {
	int device, K, nw, cu_count;
	uint64_t slots;
	void *d_ent;
	uint32_t *d_aux;
	uint64_t *d_first;
	sdt::Stats *d_stats, *h_stats;
	hipStream_t stream, copy_stream;
	uint64_t **d_idx;              // slot -> node index in the visiting order (fields of the context: load_paths reads them too)
	uint64_t *idx_slots, *idx_n;
	GraphExt **gx;
};
GraphView graph_view(sdt_ctx *c);

template <int NW> inline sdt::Table<NW> table_of(const GraphView &v)
{
	sdt::Table<NW> t;
	t.ent = (sdt::Entry<NW> *)v.d_ent;
	t.aux = v.d_aux;
	t.fslots = v.slots;
	t.first = v.d_first;
	return t;
}

inline int scan_grid(int cu_count, uint64_t items)
{
	uint64_t blocks = (items + TPB - 1) / TPB;
	// (SDT_SCAN_BLOCKS: test hook -- a handful of workgroups do all the work, so that per-wave state (the chunks of sdt_append.cuh)
	// goes through every transition on small inputs)
	static const int forced = sdt_test_env("SDT_SCAN_BLOCKS") ? atoi(sdt_test_env("SDT_SCAN_BLOCKS")) : 0;
	const uint64_t cap = forced > 0 ? (uint64_t)forced : (uint64_t)cu_count * 8;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

int sync_stats(sdt_ctx *c);        // drain + copy the device counters to h_stats (SDT_EFULL when a probe failed)
int release_pass1(sdt_ctx *c);     // pass 1 is over: give the pools of the locality pipeline back to the device
int drop_first(sdt_ctx *c);        // the visiting order is on the device: the first-occurrence ordinals are not needed any more
// large transfers between pageable host memory and the device through pinned staging buffers filled / drained by a few threads
int h2d_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes);
int d2h_big(hipStream_t copy_stream, void *dst, const void *src, size_t bytes);
// n arcs (from, to, multiplicity, first appearance) in device arrays, put in the order *.preArc lists them: from ascending, most
// recent first appearance first (sdt_gpu_graph.hip: the unit with the device-wide sorts)
int sort_arcs_for_output(hipStream_t stream, int cu_count, uint32_t *d_from, uint32_t *d_to, uint32_t *d_mult, uint64_t *d_first, uint64_t n);

}  // namespace sdti

#define hipMalloc(p, bytes) sdti::dmalloc((void **)(p), (bytes), __FILE__, __LINE__)
#define hipFree(p) sdti::dfree((p), __FILE__, __LINE__)
#define hipHostMalloc(p, bytes, flags) sdti::hmalloc((void **)(p), (bytes), (flags), __FILE__, __LINE__)
