// sdt_gpu_graph.hip -- the graph phases of pregraph on the device mirror of the graph (part of libsdt_gpu.so).
//
//   layout      sdt_gpu_layout_sorted_keys / sdt_gpu_layout_apply / sdt_gpu_export_ordered: the reference's visiting order
//               (graph.c replays only the probing of each set; sort, gather and numbering happen here)
//   dry runs    the read-only halves of removeMinorOut / removeSingleTips / removeMinorTips (cutTipPreGraph.c) and of
//               kmer2edges (node2edge.c), labelled with the connected components the ordered commits may run side by side
//               (union-find over node indices), sorted by (component, node)
// gfx950 only; no CPU fallback.  The context is seen through sdti::GraphView (sdt_internal.hpp).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "sdt_internal.hpp"

using namespace sdt;

#include "sdt_append.cuh"
#include "sdt_graph_kernels.cuh"

using sdti::fail;
using sdti::GraphView;

struct sdti::GraphExt {
	uint64_t *d_sval = nullptr;       // rank in (set, first-occurrence) order -> table slot   (layout_sorted_keys .. layout_apply)
	uint64_t n_sorted = 0;
	uint64_t *d_slot_of = nullptr;    // node index (visiting order) -> table slot
	uint64_t n_nodes = 0;
	uint64_t *d_result = nullptr;     // records of the last labelled dry run, waiting for sdt_gpu_fetch_records
	uint64_t result_words = 0;
	uint64_t result_labelled = 0;     // the first so many of them are sorted by (component, node)
	int result_stride = 0;
	uint64_t *d_wnode = nullptr;      // nodes the last sdt_gpu_minor_out_commit wrote, waiting for sdt_gpu_fetch_written
	uint32_t *d_wl = nullptr, *d_wr = nullptr;
	uint64_t n_written = 0;
	uint64_t *d_skipped = nullptr;    // records of the components that commit left to the host, waiting for sdt_gpu_fetch_skipped
	uint64_t n_skipped = 0;
	bool mo_pending = false;          // between sdt_gpu_minor_out_commit_begin and _finish: what the launched kernels work on
	uint8_t *mo_dirty = nullptr;
	unsigned long long *mo_cnt = nullptr;
	uint32_t *mo_recidx = nullptr, *mo_cstart = nullptr;
	unsigned char *d_seq = nullptr;   // bases of the edges of sdt_gpu_build_edges, waiting for sdt_gpu_fetch_edge_bases
	uint64_t seq_bytes = 0;
	uint64_t *d_pw = nullptr;         // path word of every node after sdt_gpu_build_edges (taken by sdt_gpu_load_paths)
	uint64_t pw_n = 0;
};

void sdti::graph_ext_free(GraphExt *gx)
{
	if (!gx) return;
	if (gx->d_sval) (void)hipFree(gx->d_sval);
	if (gx->d_slot_of) (void)hipFree(gx->d_slot_of);
	if (gx->d_result) (void)hipFree(gx->d_result);
	if (gx->d_wnode) (void)hipFree(gx->d_wnode);
	if (gx->d_wl) (void)hipFree(gx->d_wl);
	if (gx->d_wr) (void)hipFree(gx->d_wr);
	if (gx->d_skipped) (void)hipFree(gx->d_skipped);
	if (gx->mo_dirty) (void)hipFree(gx->mo_dirty);
	if (gx->mo_cnt) (void)hipFree(gx->mo_cnt);
	if (gx->mo_recidx) (void)hipFree(gx->mo_recidx);
	if (gx->mo_cstart) (void)hipFree(gx->mo_cstart);
	if (gx->d_seq) (void)hipFree(gx->d_seq);
	if (gx->d_pw) (void)hipFree(gx->d_pw);
	delete gx;
}

uint64_t *sdti::graph_take_path_words(GraphExt *gx, uint64_t n)
{
	if (!gx || !gx->d_pw || gx->pw_n != n) return nullptr;
	uint64_t *p = gx->d_pw;
	gx->d_pw = nullptr;
	gx->pw_n = 0;
	return p;
}

static sdti::GraphExt *ext_of(const GraphView &v)
{
	if (!*v.gx) *v.gx = new sdti::GraphExt();
	return *v.gx;
}

#define LAUNCH_NW(v, kernel, grid, ...)                                                                                      \
	do {                                                                                                                     \
		if ((v).nw == 1) hipLaunchKernelGGL(kernel<1>, dim3(grid), dim3(TPB), 0, (v).stream, sdti::table_of<1>(v), __VA_ARGS__);      \
		else if ((v).nw == 2) hipLaunchKernelGGL(kernel<2>, dim3(grid), dim3(TPB), 0, (v).stream, sdti::table_of<2>(v), __VA_ARGS__); \
		else hipLaunchKernelGGL(kernel<4>, dim3(grid), dim3(TPB), 0, (v).stream, sdti::table_of<4>(v), __VA_ARGS__);                  \
	} while (0)

// every device buffer of a call in one place: freed when the call returns, whatever the path
struct Scratch {
	void *p[96] = {};
	int n = 0;
	template <class T> hipError_t alloc(T **out, size_t bytes)
	{
		void *q = nullptr;
		if (n == (int)(sizeof p / sizeof p[0])) { *out = nullptr; return hipErrorOutOfMemory; }
		const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
		if (e == hipSuccess) p[n++] = q;
		*out = (T *)q;
		return e;
	}
	void *release(void *q)                                    // the caller keeps q
	{
		for (int i = 0; i < n; i++) if (p[i] == q) p[i] = nullptr;
		return q;
	}
	~Scratch() { for (int i = 0; i < n; i++) if (p[i]) (void)hipFree(p[i]); }
};

#define GCHK(expr) do { hipError_t e9_ = (expr); if (e9_ != hipSuccess) return fail(e9_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e9_), __FILE__, __LINE__); } while (0)

// device-wide sort of (key, value) pairs by the key bits [0, end_bit): rocPRIM's radix sort ping-pongs between the caller's input and
// output arrays (double buffers: the INPUT arrays are scratch afterwards), so its own temporary storage stays small -- with separate
// in / out arrays it asked for as much again as the pairs (10 GiB at 678 M nodes, from the driver)
template <class V>
static int sort_pairs(const GraphView &v, uint64_t *k_in, uint64_t *k_out, V *v_in, V *v_out, uint64_t n, unsigned end_bit)
{
	rocprim::double_buffer<uint64_t> dk(k_in, k_out);
	rocprim::double_buffer<V> dv(v_in, v_out);
	size_t tmp_bytes = 0;
	GCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)n, 0u, end_bit, v.stream));
	void *tmp = nullptr;
	GCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
	hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)n, 0u, end_bit, v.stream);
	if (e == hipSuccess && dk.current() != k_out) e = hipMemcpyAsync(k_out, dk.current(), n * sizeof(uint64_t), hipMemcpyDeviceToDevice, v.stream);
	if (e == hipSuccess && dv.current() != v_out) e = hipMemcpyAsync(v_out, dv.current(), n * sizeof(V), hipMemcpyDeviceToDevice, v.stream);
	const hipError_t e2 = hipStreamSynchronize(v.stream);
	(void)hipFree(tmp);
	if (e != hipSuccess || e2 != hipSuccess) return fail(SDT_EHIP, "radix sort of %llu pairs: %s", (unsigned long long)n, hipGetErrorString(e != hipSuccess ? e : e2));
	return SDT_OK;
}

template <class T>
static int exclusive_scan(const GraphView &v, const T *in, T *out, uint64_t n)
{
	size_t tmp_bytes = 0;
	GCHK(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), v.stream));
	void *tmp = nullptr;
	GCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
	const hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), v.stream);
	const hipError_t e2 = hipStreamSynchronize(v.stream);
	(void)hipFree(tmp);
	if (e != hipSuccess || e2 != hipSuccess) return fail(SDT_EHIP, "prefix sum over %llu items: %s", (unsigned long long)n, hipGetErrorString(e != hipSuccess ? e : e2));
	return SDT_OK;
}


// (from ascending, first appearance descending): stable sort by ~first, then stable sort by from
int sdti::sort_arcs_for_output(hipStream_t stream, int cu_count, uint32_t *d_from, uint32_t *d_to, uint32_t *d_mult, uint64_t *d_first, uint64_t n)
{
	if (n < 2) return SDT_OK;
	if (n >= 0xFFFFFFFFULL) return fail(SDT_ELIMIT, "%llu arcs do not fit a 32-bit permutation", (unsigned long long)n);
	Scratch S;
	uint64_t *k0, *k1, *o2;
	uint32_t *p0, *p1, *f0, *f1, *f2, *t2, *m2;
	GCHK(S.alloc(&k0, n * 8)); GCHK(S.alloc(&k1, n * 8)); GCHK(S.alloc(&o2, n * 8));
	GCHK(S.alloc(&p0, n * 4)); GCHK(S.alloc(&p1, n * 4)); GCHK(S.alloc(&f0, n * 4)); GCHK(S.alloc(&f1, n * 4));
	GCHK(S.alloc(&f2, n * 4)); GCHK(S.alloc(&t2, n * 4)); GCHK(S.alloc(&m2, n * 4));
	const dim3 grid(sdti::scan_grid(cu_count, n));
	hipLaunchKernelGGL(k_arc_keys, grid, dim3(TPB), 0, stream, d_first, n, k0, p0);
	size_t tb1 = 0, tb2 = 0;
	GCHK(rocprim::radix_sort_pairs(nullptr, tb1, k0, k1, p0, p1, (size_t)n, 0u, 64u, stream));
	GCHK(rocprim::radix_sort_pairs(nullptr, tb2, f0, f1, p1, p0, (size_t)n, 0u, 32u, stream));
	void *tmp = nullptr;
	GCHK(S.alloc(&tmp, (tb1 > tb2 ? tb1 : tb2) + 16));
	GCHK(rocprim::radix_sort_pairs(tmp, tb1, k0, k1, p0, p1, (size_t)n, 0u, 64u, stream));
	hipLaunchKernelGGL(k_arc_from_of, grid, dim3(TPB), 0, stream, d_from, p1, n, f0);
	GCHK(rocprim::radix_sort_pairs(tmp, tb2, f0, f1, p1, p0, (size_t)n, 0u, 32u, stream));
	hipLaunchKernelGGL(k_arc_gather, grid, dim3(TPB), 0, stream, p0, n, d_from, d_to, d_mult, d_first, f2, t2, m2, o2);
	GCHK(hipGetLastError());
	GCHK(hipMemcpyAsync(d_from, f2, n * 4, hipMemcpyDeviceToDevice, stream));
	GCHK(hipMemcpyAsync(d_to, t2, n * 4, hipMemcpyDeviceToDevice, stream));
	GCHK(hipMemcpyAsync(d_mult, m2, n * 4, hipMemcpyDeviceToDevice, stream));
	GCHK(hipMemcpyAsync(d_first, o2, n * 8, hipMemcpyDeviceToDevice, stream));
	GCHK(hipStreamSynchronize(stream));
	return SDT_OK;
}

// ---- records appended in chunks per wave (sdt_append.cuh): storage, and packing into one dense array ---------------------
struct ApBuf {
	uint64_t *chunks = nullptr;
	uint32_t *fill = nullptr, *off = nullptr;
	unsigned long long *cursor = nullptr;
	uint64_t cap_chunks = 0;
	int stride = 0;
};
static ApOut ap_out(const ApBuf &B) { return ApOut{B.cursor, B.cap_chunks, B.fill, nullptr}; }
// room for `records` records of `stride` words (a closed chunk holds more than AP_CH - 64 records, every wave has one open chunk)
static int ap_alloc(Scratch &S, const GraphView &v, ApBuf &B, uint64_t records, int stride)
{
	B.cap_chunks = records / (AP_CH - 64) + 1 + (uint64_t)v.cu_count * 8 * (TPB / 64);
	B.stride = stride;
	GCHK(S.alloc(&B.chunks, B.cap_chunks * AP_CH * (size_t)stride * 8));
	GCHK(S.alloc(&B.fill, (B.cap_chunks + 1) * 4)); GCHK(S.alloc(&B.off, (B.cap_chunks + 1) * 4)); GCHK(S.alloc(&B.cursor, 8));
	GCHK(hipMemsetAsync(B.fill, 0, (B.cap_chunks + 1) * 4, v.stream));
	GCHK(hipMemsetAsync(B.cursor, 0, 8, v.stream));
	return SDT_OK;
}
static void ap_free(Scratch &S, ApBuf &B)
{
	(void)hipFree(S.release(B.chunks)); (void)hipFree(S.release(B.fill)); (void)hipFree(S.release(B.off)); (void)hipFree(S.release(B.cursor));
	B = ApBuf();
}
// the records of chunks [0, n_chunks) packed in chunk order into a new array (the caller's Scratch owns it); *n_split = the records
// of the chunks before split_chunk (what an earlier kernel wrote)
static int ap_compact(Scratch &S, const GraphView &v, const ApBuf &B, uint64_t n_chunks, uint64_t split_chunk, uint64_t **out, uint64_t *n_out, uint64_t *n_split)
{
	if (n_chunks > B.cap_chunks) return fail(SDT_ESTATE, "records in chunks: %llu chunks taken, room for %llu", (unsigned long long)n_chunks, (unsigned long long)B.cap_chunks);
	const int rc = exclusive_scan<uint32_t>(v, B.fill, B.off, n_chunks + 1);          // (fill[n_chunks] is 0: off[n_chunks] = all records)
	if (rc != SDT_OK) return rc;
	uint32_t tot = 0, spl = 0;
	GCHK(hipMemcpy(&tot, B.off + n_chunks, 4, hipMemcpyDeviceToHost));
	GCHK(hipMemcpy(&spl, B.off + (split_chunk < n_chunks ? split_chunk : n_chunks), 4, hipMemcpyDeviceToHost));
	GCHK(S.alloc(out, ((size_t)tot + 1) * (size_t)B.stride * 8));
	if (n_chunks) hipLaunchKernelGGL(k_ap_compact, dim3(sdti::scan_grid(v.cu_count, n_chunks * AP_CH)), dim3(256), 0, v.stream, B.chunks, B.fill, B.off,
	                                 (unsigned long long)n_chunks, B.stride, *out);
	GCHK(hipGetLastError());
	*n_out = tot;
	if (n_split) *n_split = spl;
	return SDT_OK;
}

// ---- the whole layout on the device: sort, replay of the probing (fixed point per growth), numbering ---------------------
#include <math.h>
#include <time.h>
#include <vector>
static int rp_prime(uint64_t num)                    // find_next_prime_kh's test (newhash.c:116-158): the bound is (ubyte8)sqrt((float)n)
{
	if (num < 4) return 1;
	if (num % 2 == 0) return 0;
	const uint64_t lim = (uint64_t)sqrt((float)num);
	for (uint64_t i = 3; i < lim; i += 2)
		if (num % i == 0) return 0;
	return 1;
}
static uint64_t rp_next_prime(uint64_t n) { if (n % 2 == 0) n++; while (!rp_prime(n)) n += 2; return n; }
static uint64_t rp_next_size(uint64_t size, double lf, uint64_t count)      // encap_kmerset's growth (newhash.c:318-330)
{
	uint64_t n = size;
	do {
		n = n < 0xFFFFFFFu ? n << 1 : n + 0xFFFFFFu;
		n = rp_next_prime(n);
	} while (n * lf < (double)(count + 1));
	return n;
}

extern "C" int sdt_gpu_layout_on_device(sdt_ctx *c, int p, int nw_variant, int small_init, uint64_t *set_start, uint64_t *n_out)
{
	if (!c || !n_out || !set_start) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v0 = sdti::graph_view(c);
	HIPCHK(hipSetDevice(v0.device));
	struct timespec ts0_; clock_gettime(CLOCK_MONOTONIC, &ts0_);
	const double t_call = ts0_.tv_sec * 1e3 + ts0_.tv_nsec * 1e-6;
	const bool timing = sdt_env("SDT_TIMING") != nullptr;
	auto tick = [&](const char *what) {              // (SDT_TIMING: where the call's time goes, on stderr)
		if (!timing) return;
		struct timespec t_;
		clock_gettime(CLOCK_MONOTONIC, &t_);
		fprintf(stderr, "[device]       %s at %.1f ms\n", what, t_.tv_sec * 1e3 + t_.tv_nsec * 1e-6 - t_call);
	};
	int rc = sdti::release_pass1(c);
	if (rc != SDT_OK) return rc;
	tick("pass 1 released");
	const GraphView v = sdti::graph_view(c);
	const uint64_t n = v.h_stats->distinct;
	*n_out = n;
	if (!v.d_first) return fail(SDT_ESTATE, "first-occurrence ordinals were not tracked: init with SDT_FLAG_TRACK_FIRST");
	if (p < 1 || p > 256) return fail(SDT_EINVAL, "layout on the device: 1..256 sets, asked for %d", p);
	if (nw_variant < v.nw || nw_variant > 4) return fail(SDT_EINVAL, "a %d-word variant cannot hold %d-word keys", nw_variant, v.nw);
	if (n >= 0xFFFFFFF0ULL) return fail(SDT_EINVAL, "layout on the device: 32-bit ranks, %llu nodes", (unsigned long long)n);
	sdti::GraphExt *gx = ext_of(v);
	if (gx->d_sval) { (void)hipFree(gx->d_sval); gx->d_sval = nullptr; gx->n_sorted = 0; }
	Scratch S;
	const uint64_t m = n ? n : 1;
	// ---- sort by (set, first occurrence), keys gathered in that order (as sdt_gpu_layout_sorted_keys)
	uint64_t *k0, *k1, *v0s, *v1s, *d_keys, *d_ss;
	unsigned long long *d_cur;
	GCHK(S.alloc(&k0, m * 8)); GCHK(S.alloc(&k1, m * 8)); GCHK(S.alloc(&v0s, m * 8)); GCHK(S.alloc(&v1s, m * 8));
	GCHK(S.alloc(&d_cur, 8)); GCHK(S.alloc(&d_ss, (size_t)(p + 1) * 8));
	GCHK(hipMemsetAsync(d_cur, 0, 8, v.stream));
	const int g = sdti::scan_grid(v.cu_count, v.slots);
	LAUNCH_NW(v, k_layout_keys, g, (uint32_t)p, nw_variant, k0, v0s, (unsigned long long)n, d_cur, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "layout: %llu nodes without a usable first-occurrence ordinal", (unsigned long long)v.h_stats->probe_fail);
	tick("sort keys made");
	unsigned end_bit = 56;
	for (int q = p - 1; q > 0; q >>= 1) end_bit++;
	rc = sort_pairs<uint64_t>(v, k0, k1, v0s, v1s, n, end_bit);
	if (rc != SDT_OK) return rc;
	tick("sorted");
	hipLaunchKernelGGL(k_layout_set_starts, dim3((p + 1 + 63) / 64), dim3(64), 0, v.stream, k1, n, (uint32_t)p, d_ss);
	GCHK(hipGetLastError());
	GCHK(hipMemcpyAsync(set_start, d_ss, (size_t)(p + 1) * 8, hipMemcpyDeviceToHost, v.stream));
	if (v.nw == 1) d_keys = k0; else GCHK(S.alloc(&d_keys, m * v.nw * 8));
	LAUNCH_NW(v, k_layout_gather_keys, sdti::scan_grid(v.cu_count, m), v1s, n, d_keys);
	GCHK(hipGetLastError());
	GCHK(hipStreamSynchronize(v.stream));
	tick("keys gathered");
	(void)hipFree(S.release(k1));                             // (the sorted sort keys are not needed any more)
	if (v.nw != 1) (void)hipFree(S.release(k0));
	(void)hipFree(S.release(v0s));
	// ---- every set's growth schedule (a function of its number of keys only)
	struct Gen { uint32_t lo, hi, size; };                   // ids [lo, hi) are put into a table of `size` slots (after a growth from the previous size)
	std::vector<std::vector<Gen>> sched(p);
	std::vector<RpSet> sets(p);
	const double lf = (double)0.77f;
	uint64_t tab_total = 0, max_size = 0;
	size_t max_gens = 0;
	for (int s = 0; s < p; s++) {
		const uint64_t ms = set_start[s + 1] - set_start[s];
		uint64_t size = (nw_variant != 1 && small_init) ? 3 : rp_next_prime(1024);      // init_kmerset (prlHashReads.c:402-423, newhash.c:163-166)
		uint64_t max = (uint64_t)(size * 0.77f), lo = 0;
		for (;;) {
			const uint64_t hi = ms < max ? ms : max;
			sched[s].push_back(Gen{(uint32_t)lo, (uint32_t)hi, (uint32_t)size});
			if (hi >= ms) break;
			size = rp_next_size(size, lf, max);                // the put of id == max finds count + 1 > max
			if (size >= 0xFFFFFFFFULL) return fail(SDT_EINVAL, "layout on the device: a set's table passes 2^32 slots");
			lo = hi;
			max = (uint64_t)(size * lf);
		}
		sets[s].key0 = set_start[s]; sets[s].tab0 = tab_total; sets[s].m = (uint32_t)ms;
		tab_total += size;
		if (size > max_size) max_size = size;
		if (sched[s].size() > max_gens) max_gens = sched[s].size();
	}
	// table word of a growth's rounds: time << qbits | q + 1 (q = old slot, time = old slot << RP_DEPTH_BITS | depth); a dirty cluster's
	// list entry keeps the set in 10 bits
	int qbits = 1;
	while ((1ULL << qbits) <= max_size) qbits++;
	if (2 * qbits + RP_DEPTH_BITS > 64 || p >= 1024) return fail(SDT_ELIMIT, "layout on the device: twice %d slot bits do not fit the table word (or %d sets the list entry)", qbits, p);
	tick("schedule made");
	// ---- buffers
	unsigned long long *tab[2], *d_time, *d_saved, *d_list[2], *d_pre;
	uint32_t *d_home, *d_occ, *d_rank;
	RpSet *d_sets;
	unsigned int *d_flags;
	RpRound *d_round;
	GCHK(S.alloc(&tab[0], (tab_total + 1) * 8)); GCHK(S.alloc(&tab[1], (tab_total + 1) * 8));
	uint32_t *d_home_slot;
	// per OLD slot of a growth: time and home; per new slot: the word a round took out of it; per entry: the two lists of the rounds
	// (an entry whose time changed is listed once)
	// (sdt_append.cuh: the lists are written in chunks of AP_CH entries per wave, the open chunk of every wave has unused slots)
	// A round lists the entries whose time changed: a fifth of the entries of a growth in round 0 (tools/replay_fixed_point.c), i.e. an
	// eighth of all keys when the largest set grows; room for a quarter of all keys (past that: SDT_ELIMIT, the caller replays on the host)
	const unsigned long long list_chunks = (m / 4 + 1) / (AP_CH - 64) + 1 + (unsigned long long)v.cu_count * 8 * (TPB / 64), list_cap = list_chunks * AP_CH;
	GCHK(S.alloc(&d_time, (tab_total + 1) * 8)); GCHK(S.alloc(&d_home_slot, (tab_total + 1) * 4)); GCHK(S.alloc(&d_saved, (tab_total + 1) * 8));
	GCHK(S.alloc(&d_list[0], list_cap * 8)); GCHK(S.alloc(&d_list[1], list_cap * 8)); GCHK(S.alloc(&d_round, sizeof(RpRound)));
	GCHK(S.alloc(&d_home, m * 4)); GCHK(S.alloc(&d_sets, (size_t)p * sizeof(RpSet))); GCHK(S.alloc(&d_pre, (size_t)(p + 1) * 8)); GCHK(S.alloc(&d_flags, 4));
	GCHK(hipMemsetAsync(tab[0], 0, (tab_total + 1) * 8, v.stream));
	tick("buffers made");
	std::vector<unsigned long long> pre(p + 1);
	auto upload = [&](void) -> int {
		GCHK(hipStreamSynchronize(v.stream));                  // (kernels in flight read the old copies; the host vectors change right after)
		GCHK(hipMemcpy(d_sets, sets.data(), (size_t)p * sizeof(RpSet), hipMemcpyHostToDevice));
		GCHK(hipMemcpy(d_pre, pre.data(), (size_t)(p + 1) * 8, hipMemcpyHostToDevice));
		return SDT_OK;
	};
	auto grid = [&](unsigned long long items) { return dim3(sdti::scan_grid(v.cu_count, items ? items : 1)); };
	unsigned int h_flags = 0;
	int cur = 0, total_rounds = 0;
	auto now_ms = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
	const double t_start = now_ms();
	double t_rehash = 0, t_put = 0, t_strip = 0;
	if (timing) fprintf(stderr, "[device]     layout: release + sort + gather + schedule + buffers %.1f ms\n", t_start - t_call);
	for (size_t gen = 0; gen < max_gens; gen++) {
		// ---- growth into this generation's size (every set that has a generation `gen`, except the first)
		if (gen > 0) {
			unsigned long long old_total = 0, cnt_total = 0;
			for (int s = 0; s < p; s++) {
				const bool on = gen < sched[s].size();
				sets[s].old_size = on ? sched[s][gen - 1].size : 0;
				sets[s].size = on ? sched[s][gen].size : sets[s].size;
				sets[s].lo = 0; sets[s].hi = on ? sched[s][gen].lo : 0;            // ids [0, lo) are in the table
				pre[s] = cnt_total;
				cnt_total += sets[s].hi;
			}
			(void)cnt_total;
			for (int s = 0; s < p; s++) { pre[s] = old_total; old_total += sets[s].old_size; }
			pre[p] = old_total;
			rc = upload(); if (rc != SDT_OK) return rc;
			const int nxt = cur ^ 1;
			const double t_g0 = now_ms();
			const int rounds0 = total_rounds;
			// time (q, 0) and the home in the new geometry, per old slot
			if (v.nw == 1) hipLaunchKernelGGL(k_rp_rehash_init<1>, grid(old_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, tab[cur], d_time, d_home_slot);
			else if (v.nw == 2) hipLaunchKernelGGL(k_rp_rehash_init<2>, grid(old_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, tab[cur], d_time, d_home_slot);
			else hipLaunchKernelGGL(k_rp_rehash_init<4>, grid(old_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, tab[cur], d_time, d_home_slot);
			for (int s = 0; s < p; s++)                           // (only the regions of the sets that grow, at their new size: the early growths are tiny)
				if (sets[s].old_size) GCHK(hipMemsetAsync(tab[nxt] + sets[s].tab0, 0, (size_t)sets[s].size * 8, v.stream));
			GCHK(hipMemsetAsync(d_round, 0, sizeof(RpRound), v.stream));
			// round 0: everybody
			hipLaunchKernelGGL(k_rp_ins_all, grid(old_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, qbits, tab[cur], tab[nxt], d_home_slot, d_time, d_round);
			hipLaunchKernelGGL(k_rp_eval_all, grid(old_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, qbits, tab[cur], tab[nxt], d_home_slot, d_time,
			                   d_list[0], list_chunks, d_round);
			GCHK(hipGetLastError());
			// (SDT_RP_MAX_ROUNDS: test hook -- a cap that real data passes, so that the caller's other path is exercised)
			static const int max_rounds = sdt_test_env("SDT_RP_MAX_ROUNDS") ? atoi(sdt_test_env("SDT_RP_MAX_ROUNDS")) : 60;
			int lc = 0;
			for (int round = 0;; round++) {
				RpRound h_round;
				GCHK(hipMemcpyAsync(&h_round, d_round, sizeof(RpRound), hipMemcpyDeviceToHost, v.stream));
				GCHK(hipStreamSynchronize(v.stream));
				total_rounds++;
				if (h_round.flags) return fail(SDT_ELIMIT, "layout on the device: %s", (h_round.flags & 2u) ? "an insertion found no slot" :
				                               (h_round.flags & 4u) ? "an eviction chain or a stretch of slots past its field" : "the list of a round is full");
				if (!h_round.n_next) break;
				if (round >= max_rounds) return fail(SDT_ELIMIT, "layout on the device: a growth did not settle in %d rounds", max_rounds);
				// from the home of every changed entry to the end of its cluster: taken out, laid out again, the old slots in there evaluated again
				const unsigned long long n_list = h_round.n_next * AP_CH;
				GCHK(hipMemsetAsync(d_round, 0, sizeof(RpRound), v.stream));
				hipLaunchKernelGGL(k_rp_collect, grid(n_list), dim3(TPB), 0, v.stream, d_sets, tab[nxt], d_saved, d_list[lc], n_list, d_round);
				hipLaunchKernelGGL(k_rp_ins_list, grid(n_list), dim3(TPB), 0, v.stream, d_sets, d_pre, qbits, tab[nxt], d_saved, d_home_slot, d_time, d_list[lc], n_list, d_round);
				hipLaunchKernelGGL(k_rp_eval_list, grid(n_list), dim3(TPB), 0, v.stream, d_sets, d_pre, qbits, tab[cur], tab[nxt], d_home_slot, d_time,
				                   d_list[lc], n_list, d_list[lc ^ 1], list_chunks, d_round);
				GCHK(hipGetLastError());
				lc ^= 1;
			}
			const double t_g1 = now_ms();
			t_rehash += t_g1 - t_g0;
			if (timing && old_total > (1u << 24)) fprintf(stderr, "[device]     growth %zu: %llu old slots, %d rounds, %.1f ms\n", gen, old_total, total_rounds - rounds0, t_g1 - t_g0);
			// the settled layout without its times is the table of this generation
			unsigned long long new_total = 0;
			for (int s = 0; s < p; s++) { pre[s] = new_total; new_total += gen < sched[s].size() ? sets[s].size : 0; }
			pre[p] = new_total;
			// (sets without this generation keep their table: copy their region over unchanged)
			for (int s = 0; s < p; s++)
				if (gen >= sched[s].size())
					GCHK(hipMemcpyAsync(tab[nxt] + sets[s].tab0, tab[cur] + sets[s].tab0, (size_t)sets[s].size * 8, hipMemcpyDeviceToDevice, v.stream));
			rc = upload(); if (rc != SDT_OK) return rc;
			{
				// slots of the sets that grew: pre counts only those (others have size 0 in the prefix)
				std::vector<RpSet> grown = sets;
				for (int s = 0; s < p; s++) if (gen >= sched[s].size()) grown[s].size = 0;
				GCHK(hipMemcpy(d_sets, grown.data(), (size_t)p * sizeof(RpSet), hipMemcpyHostToDevice));
				hipLaunchKernelGGL(k_rp_slots, grid(new_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, 0, qbits, tab[cur], tab[nxt], (uint32_t *)nullptr);
				GCHK(hipGetLastError());
				GCHK(hipStreamSynchronize(v.stream));
			}
			cur = nxt;
			t_strip += now_ms() - t_g1;
		}
		// ---- the puts of this generation
		const double t_p0 = now_ms();
		unsigned long long put_total = 0;
		for (int s = 0; s < p; s++) {
			const bool on = gen < sched[s].size();
			sets[s].lo = on ? sched[s][gen].lo : 0;
			sets[s].hi = on ? sched[s][gen].hi : 0;
			if (on) sets[s].size = sched[s][gen].size;
			pre[s] = put_total;
			put_total += sets[s].hi - sets[s].lo;
		}
		pre[p] = put_total;
		rc = upload(); if (rc != SDT_OK) return rc;
		GCHK(hipMemsetAsync(d_flags, 0, 4, v.stream));
		if (v.nw == 1) hipLaunchKernelGGL(k_rp_home<1>, grid(put_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, 0, d_home);
		else if (v.nw == 2) hipLaunchKernelGGL(k_rp_home<2>, grid(put_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, 0, d_home);
		else hipLaunchKernelGGL(k_rp_home<4>, grid(put_total), dim3(TPB), 0, v.stream, d_keys, d_sets, d_pre, p, 0, d_home);
		hipLaunchKernelGGL(k_rp_put, grid(put_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, d_home, tab[cur], d_flags);
		GCHK(hipGetLastError());
		GCHK(hipMemcpyAsync(&h_flags, d_flags, 4, hipMemcpyDeviceToHost, v.stream));
		GCHK(hipStreamSynchronize(v.stream));
		if (h_flags) return fail(SDT_ELIMIT, "layout on the device: a put found no slot");
		t_put += now_ms() - t_p0;
	}
	if (timing) fprintf(stderr, "[device]     layout replay so far %.1f ms: rehash rounds %.1f, strip + copies %.1f, puts %.1f ms\n", now_ms() - t_start, t_rehash, t_strip, t_put);
	// ---- the visiting order: the sets one after the other, every set's slots in order
	unsigned long long slot_total = 0;
	for (int s = 0; s < p; s++) { sets[s].size = sched[s].back().size; pre[s] = slot_total; slot_total += sets[s].size; }
	pre[p] = slot_total;
	rc = upload(); if (rc != SDT_OK) return rc;
	// (the buffers of the replay that are free now make room for the flags and their prefix sum)
	(void)hipFree(S.release(tab[cur ^ 1])); (void)hipFree(S.release(d_time)); (void)hipFree(S.release(d_home_slot));
	(void)hipFree(S.release(d_saved)); (void)hipFree(S.release(d_list[0])); (void)hipFree(S.release(d_list[1]));
	uint64_t *d_order;
	GCHK(S.alloc(&d_occ, (slot_total + 1) * 4)); GCHK(S.alloc(&d_rank, (slot_total + 1) * 4)); GCHK(S.alloc(&d_order, m * 8));
	hipLaunchKernelGGL(k_rp_slots, grid(slot_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, 1, qbits, (const unsigned long long *)nullptr, tab[cur], d_occ);
	GCHK(hipGetLastError());
	rc = exclusive_scan<uint32_t>(v, d_occ, d_rank, slot_total);
	if (rc != SDT_OK) return rc;
	hipLaunchKernelGGL(k_rp_order, grid(slot_total), dim3(TPB), 0, v.stream, d_sets, d_pre, p, tab[cur], d_occ, d_rank, d_order);
	GCHK(hipGetLastError());
	GCHK(hipStreamSynchronize(v.stream));
	(void)hipFree(S.release(tab[cur])); (void)hipFree(S.release(d_occ)); (void)hipFree(S.release(d_rank)); (void)hipFree(S.release(d_home));
	if (timing) fprintf(stderr, "[device]     layout: order extracted at %.1f ms\n", now_ms() - t_call);
	// ---- number the nodes (as sdt_gpu_layout_apply)
	if (*v.d_idx) { (void)hipFree(*v.d_idx); *v.d_idx = nullptr; }
	*v.idx_slots = *v.idx_n = 0;
	if (gx->d_slot_of) { (void)hipFree(gx->d_slot_of); gx->d_slot_of = nullptr; gx->n_nodes = 0; }
	// the replay is through (every limit it can run into lies behind): the first-occurrence ordinals have done their work, and the
	// node index is exactly as large -- 8 bytes per table slot that it finds in the arena instead of asking the driver
	if (!sdt_test_env("SDT_KEEP_FIRST")) { rc = sdti::drop_first(c); if (rc != SDT_OK) return rc; }
	uint64_t *d_idx, *d_slot_of;
	GCHK(S.alloc(&d_idx, v.slots * 8)); GCHK(S.alloc(&d_slot_of, m * 8));
	GCHK(hipMemsetAsync(d_idx, 0xFF, v.slots * 8, v.stream));
	hipLaunchKernelGGL(k_layout_apply, dim3(sdti::scan_grid(v.cu_count, m)), dim3(TPB), 0, v.stream, v1s, d_order, n, d_idx, d_slot_of, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "layout on the device: %llu positions of the order are not ranks", (unsigned long long)v.h_stats->probe_fail);
	*v.d_idx = (uint64_t *)S.release(d_idx);
	*v.idx_slots = v.slots;
	*v.idx_n = n;
	gx->d_slot_of = (uint64_t *)S.release(d_slot_of);
	gx->n_nodes = n;
	if (timing) fprintf(stderr, "[device]     layout: whole call %.1f ms\n", now_ms() - t_call);
	if (sdt_env("SDT_TIMING")) fprintf(stderr, "[device]   layout replay: %zu generations, %d rounds of timed insertion in all, %llu table slots\n", max_gens, total_rounds, (unsigned long long)tab_total);
	return SDT_OK;
}

extern "C" {

// ---- layout ---------------------------------------------------------------------------------------------------------
int sdt_gpu_layout_sorted_keys(sdt_ctx *c, int p, int nw_variant, uint64_t *keys, uint64_t max_nodes, uint64_t *set_start, uint64_t *n_out)
{
	if (!c || !n_out) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v0 = sdti::graph_view(c);
	HIPCHK(hipSetDevice(v0.device));
	int rc = sdti::release_pass1(c);                          // drains pass 1; its pools are not needed any more
	if (rc != SDT_OK) return rc;
	const GraphView v = sdti::graph_view(c);                  // (the drain may have grown the table)
	const uint64_t n = v.h_stats->distinct;
	*n_out = n;
	if (!keys && !set_start) return SDT_OK;
	if (!keys || !set_start) return fail(SDT_EINVAL, "NULL argument");
	if (!v.d_first) return fail(SDT_ESTATE, "first-occurrence ordinals were not tracked: init with SDT_FLAG_TRACK_FIRST");
	if (p < 1 || p > 256) return fail(SDT_EINVAL, "layout on the device: 1..256 sets, asked for %d", p);
	if (nw_variant < v.nw || nw_variant > 4) return fail(SDT_EINVAL, "a %d-word variant cannot hold %d-word keys", nw_variant, v.nw);
	if (max_nodes < n) return fail(SDT_EINVAL, "key array holds %llu nodes, the table has %llu", (unsigned long long)max_nodes, (unsigned long long)n);
	sdti::GraphExt *gx = ext_of(v);
	if (gx->d_sval) { (void)hipFree(gx->d_sval); gx->d_sval = nullptr; gx->n_sorted = 0; }
	Scratch S;
	const uint64_t m = n ? n : 1;
	uint64_t *k0, *k1, *v0s, *v1s, *d_keys, *d_ss;
	unsigned long long *d_cur;
	GCHK(S.alloc(&k0, m * 8)); GCHK(S.alloc(&k1, m * 8)); GCHK(S.alloc(&v0s, m * 8)); GCHK(S.alloc(&v1s, m * 8));
	GCHK(S.alloc(&d_cur, 8)); GCHK(S.alloc(&d_ss, (size_t)(p + 1) * 8));
	GCHK(hipMemsetAsync(d_cur, 0, 8, v.stream));
	const int g = sdti::scan_grid(v.cu_count, v.slots);
	LAUNCH_NW(v, k_layout_keys, g, (uint32_t)p, nw_variant, k0, v0s, (unsigned long long)n, d_cur, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "layout: %llu nodes without a usable first-occurrence ordinal (>= 2^56, or more nodes than counted)", (unsigned long long)v.h_stats->probe_fail);
	unsigned end_bit = 56;
	for (int q = p - 1; q > 0; q >>= 1) end_bit++;
	rc = sort_pairs<uint64_t>(v, k0, k1, v0s, v1s, n, end_bit);
	if (rc != SDT_OK) return rc;
	hipLaunchKernelGGL(k_layout_set_starts, dim3((p + 1 + 63) / 64), dim3(64), 0, v.stream, k1, n, (uint32_t)p, d_ss);
	GCHK(hipGetLastError());
	GCHK(hipMemcpyAsync(set_start, d_ss, (size_t)(p + 1) * 8, hipMemcpyDeviceToHost, v.stream));
	// k0 is free again: the keys in sorted order go there (NW words each: reuse needs NW * n words)
	if (v.nw == 1) d_keys = k0; else GCHK(S.alloc(&d_keys, m * v.nw * 8));
	LAUNCH_NW(v, k_layout_gather_keys, sdti::scan_grid(v.cu_count, m), v1s, n, d_keys);
	GCHK(hipGetLastError());
	GCHK(hipStreamSynchronize(v.stream));
	rc = sdti::d2h_big(v.copy_stream, keys, d_keys, n * v.nw * 8);
	if (rc != SDT_OK) return rc;
	gx->d_sval = (uint64_t *)S.release(v1s);
	gx->n_sorted = n;
	return SDT_OK;
}

int sdt_gpu_layout_apply(sdt_ctx *c, const uint64_t *order, uint64_t n)
{
	if (!c || (n && !order)) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->d_sval || gx->n_sorted != n) return fail(SDT_ESTATE, "call sdt_gpu_layout_sorted_keys first (it sorted %llu nodes, the order has %llu)", (unsigned long long)gx->n_sorted, (unsigned long long)n);
	HIPCHK(hipSetDevice(v.device));
	if (*v.d_idx) { (void)hipFree(*v.d_idx); *v.d_idx = nullptr; }
	*v.idx_slots = *v.idx_n = 0;
	if (gx->d_slot_of) { (void)hipFree(gx->d_slot_of); gx->d_slot_of = nullptr; gx->n_nodes = 0; }
	Scratch S;
	uint64_t *d_order, *d_idx, *d_slot_of;
	const uint64_t m = n ? n : 1;
	GCHK(S.alloc(&d_order, m * 8)); GCHK(S.alloc(&d_idx, v.slots * 8)); GCHK(S.alloc(&d_slot_of, m * 8));
	GCHK(hipMemsetAsync(d_idx, 0xFF, v.slots * 8, v.stream));
	int rc = sdti::h2d_big(v.copy_stream, d_order, order, n * 8);
	if (rc != SDT_OK) return rc;
	hipLaunchKernelGGL(k_layout_apply, dim3(sdti::scan_grid(v.cu_count, m)), dim3(TPB), 0, v.stream, gx->d_sval, d_order, n, d_idx, d_slot_of, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_EINVAL, "layout order: %llu entries are not ranks below %llu", (unsigned long long)v.h_stats->probe_fail, (unsigned long long)n);
	*v.d_idx = (uint64_t *)S.release(d_idx);
	*v.idx_slots = v.slots;
	*v.idx_n = n;
	gx->d_slot_of = (uint64_t *)S.release(d_slot_of);
	gx->n_nodes = n;
	(void)hipFree(gx->d_sval);
	gx->d_sval = nullptr;
	gx->n_sorted = 0;
	return SDT_OK;
}

int sdt_gpu_export_ordered(sdt_ctx *c, uint64_t *keys, uint32_t *l_links, uint32_t *r_flags, uint32_t *count, uint64_t n)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->d_slot_of || gx->n_nodes != n) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply first (it numbered %llu nodes, the arrays hold %llu)", (unsigned long long)gx->n_nodes, (unsigned long long)n);
	HIPCHK(hipSetDevice(v.device));
	// in pieces of 32 M nodes, double buffered: the kernel of piece k + 1 runs while piece k is on the link
	const uint64_t PIECE = (uint64_t)32 << 20;
	Scratch S;
	uint64_t *dk[2] = {};
	uint32_t *dl[2] = {}, *dr[2] = {}, *dc[2] = {};
	hipEvent_t ev[2] = {};
	for (int b = 0; b < 2; b++) {
		if (keys) GCHK(S.alloc(&dk[b], PIECE * v.nw * 8));
		if (l_links) GCHK(S.alloc(&dl[b], PIECE * 4));
		if (r_flags) GCHK(S.alloc(&dr[b], PIECE * 4));
		if (count) GCHK(S.alloc(&dc[b], PIECE * 4));
		GCHK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
	}
	int rc = SDT_OK;
	uint64_t prev0 = 0, prevn = 0;
	int prevb = -1;
	for (uint64_t v0 = 0, k = 0; rc == SDT_OK && (v0 < n || prevb >= 0); v0 += PIECE, k++) {
		const int b = (int)(k & 1);
		const uint64_t cnt = v0 < n ? (n - v0 < PIECE ? n - v0 : PIECE) : 0;
		if (cnt) {
			LAUNCH_NW(v, k_export_ordered, sdti::scan_grid(v.cu_count, cnt), gx->d_slot_of, v0, cnt, dk[b], dl[b], dr[b], dc[b]);
			if (hipGetLastError() != hipSuccess || hipEventRecord(ev[b], v.stream) != hipSuccess) { rc = fail(SDT_EHIP, "export: launch failed"); break; }
		}
		if (prevb >= 0) {
			if (hipEventSynchronize(ev[prevb]) != hipSuccess) { rc = fail(SDT_EHIP, "export: event wait failed"); break; }
			if (keys && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, keys + prev0 * v.nw, dk[prevb], prevn * v.nw * 8);
			if (l_links && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, l_links + prev0, dl[prevb], prevn * 4);
			if (r_flags && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, r_flags + prev0, dr[prevb], prevn * 4);
			if (count && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, count + prev0, dc[prevb], prevn * 4);
		}
		prevb = cnt ? b : -1;
		prev0 = v0;
		prevn = cnt;
	}
	for (int b = 0; b < 2; b++) if (ev[b]) (void)hipEventDestroy(ev[b]);
	return rc;
}

int sdt_gpu_update_nodes_by_index(sdt_ctx *c, const uint64_t *node, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n)
{
	if (!c || (n && (!node || !l_links || !r_flags))) return fail(SDT_EINVAL, "NULL argument");
	if (!n) return SDT_OK;
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->d_slot_of) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply first");
	HIPCHK(hipSetDevice(v.device));
	Scratch S;
	uint64_t *d_n;
	uint32_t *d_l, *d_r;
	GCHK(S.alloc(&d_n, n * 8)); GCHK(S.alloc(&d_l, n * 4)); GCHK(S.alloc(&d_r, n * 4));
	int rc = sdti::h2d_big(v.copy_stream, d_n, node, n * 8);
	if (rc == SDT_OK) rc = sdti::h2d_big(v.copy_stream, d_l, l_links, n * 4);
	if (rc == SDT_OK) rc = sdti::h2d_big(v.copy_stream, d_r, r_flags, n * 4);
	if (rc != SDT_OK) return rc;
	LAUNCH_NW(v, k_update_by_index, sdti::scan_grid(v.cu_count, n), gx->d_slot_of, gx->n_nodes, d_n, d_l, d_r, n, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_EINVAL, "sdt_gpu_update_nodes_by_index: %llu indices past the last node", (unsigned long long)v.h_stats->probe_fail);
	return SDT_OK;
}

// ---- labelled dry runs ------------------------------------------------------------------------------------------------
// label the first n_label records (stride words each, node in the low 56 bits of word 0, label into word stride - 1) with the
// roots of `parent`, sort them by (label, node) and leave all n records in gx->d_result
static int label_sort_keep(sdt_ctx *c, const GraphView &v, uint32_t *parent, uint64_t *d_rec, uint64_t n, uint64_t n_label, int stride)
{
	sdti::GraphExt *gx = ext_of(v);
	if (gx->d_result) { (void)hipFree(gx->d_result); gx->d_result = nullptr; gx->result_words = 0; }
	Scratch S;
	uint64_t *k0, *k1, *d_out;
	uint32_t *p0, *p1;
	const uint64_t m = n_label ? n_label : 1;
	GCHK(S.alloc(&k0, m * 8)); GCHK(S.alloc(&k1, m * 8)); GCHK(S.alloc(&p0, m * 4)); GCHK(S.alloc(&p1, m * 4));
	GCHK(S.alloc(&d_out, (n ? n : 1) * (size_t)stride * 8));
	if (n_label) {
		hipLaunchKernelGGL(k_uf_label, dim3(sdti::scan_grid(v.cu_count, n_label)), dim3(TPB), 0, v.stream, parent, d_rec, n_label, stride, stride - 1, k0, p0);
		GCHK(hipGetLastError());
		const int rc = sort_pairs<uint32_t>(v, k0, k1, p0, p1, n_label, 64);
		if (rc != SDT_OK) return rc;
		hipLaunchKernelGGL(k_gather_records, dim3(sdti::scan_grid(v.cu_count, n_label * stride)), dim3(TPB), 0, v.stream, d_rec, p1, n_label, stride, d_out);
		GCHK(hipGetLastError());
	}
	if (n > n_label) {                                        // (the records behind the sorted ones keep their order; they get their label too)
		hipLaunchKernelGGL(k_uf_label_only, dim3(sdti::scan_grid(v.cu_count, n - n_label)), dim3(TPB), 0, v.stream, parent, d_rec, n_label, n, stride, stride - 1);
		GCHK(hipGetLastError());
		GCHK(hipMemcpyAsync(d_out + n_label * stride, d_rec + n_label * stride, (n - n_label) * (size_t)stride * 8, hipMemcpyDeviceToDevice, v.stream));
	}
	GCHK(hipStreamSynchronize(v.stream));
	(void)c;
	gx->d_result = (uint64_t *)S.release(d_out);
	gx->result_words = n * (uint64_t)stride;
	gx->result_labelled = n_label;
	gx->result_stride = stride;
	return SDT_OK;
}

int sdt_gpu_tip_walks_labelled(sdt_ctx *c, int thin, int cut_len, uint64_t *n_records)
{
	if (!c || !n_records) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	if (!*v.d_idx || *v.idx_slots != v.slots) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply (or sdt_gpu_set_node_index) first");
	const uint64_t nn = *v.idx_n;
	if (nn >= 0xFFFFFFF0ULL) return fail(SDT_EINVAL, "component labels are 32-bit node indices: %llu nodes", (unsigned long long)nn);
	HIPCHK(hipSetDevice(v.device));
	Scratch S;
	uint32_t *parent;
	unsigned long long *d_cur, h = 0;
	GCHK(S.alloc(&parent, (nn + 1) * 4)); GCHK(S.alloc(&d_cur, 8));
	hipLaunchKernelGGL(k_uf_init, dim3(sdti::scan_grid(v.cu_count, nn + 1)), dim3(TPB), 0, v.stream, parent, nn + 1);
	const int g = sdti::scan_grid(v.cu_count, v.slots);
	uint64_t cap = nn / 8 + 4096;
	uint64_t *d_rec = nullptr;
	ApBuf B;
	for (int attempt = 0; attempt < 2; attempt++) {
		// the dead ends first (a list of slots), then one lane per walk
		unsigned long long *d_list, *d_lcur, n_lchunks = 0;
		const unsigned long long l_chunks = cap / (AP_CH - 64) + 1 + (unsigned long long)v.cu_count * 8 * (TPB / 64);
		GCHK(S.alloc(&d_list, l_chunks * AP_CH * 8)); GCHK(S.alloc(&d_lcur, 8));
		GCHK(hipMemsetAsync(d_lcur, 0, 8, v.stream));
		int rc = ap_alloc(S, v, B, cap, 3);
		if (rc != SDT_OK) return rc;
		LAUNCH_NW(v, k_tip_starts, g, thin, d_list, ApOut{d_lcur, l_chunks, nullptr, d_list});
		GCHK(hipGetLastError());
		GCHK(hipMemcpyAsync(&n_lchunks, d_lcur, 8, hipMemcpyDeviceToHost, v.stream));
		GCHK(hipStreamSynchronize(v.stream));
		if (n_lchunks <= l_chunks) {
			const unsigned long long n_list = n_lchunks * AP_CH;
			LAUNCH_NW(v, k_tip_walks_list, sdti::scan_grid(v.cu_count, n_list), *v.d_idx, v.K, thin, cut_len, d_list, n_list, v.d_stats, B.chunks, 3, ap_out(B));
			GCHK(hipGetLastError());
			GCHK(hipMemcpyAsync(&h, B.cursor, 8, hipMemcpyDeviceToHost, v.stream));
			rc = sdti::sync_stats(c);
			if (rc != SDT_OK) return fail(SDT_ESTATE, "sdt_gpu_tip_walks_labelled: %llu walks left the graph", (unsigned long long)v.h_stats->probe_fail);
		}
		(void)hipFree(S.release(d_list)); (void)hipFree(S.release(d_lcur));
		if (n_lchunks <= l_chunks && h <= B.cap_chunks) break;
		if (attempt) return fail(SDT_ESTATE, "sdt_gpu_tip_walks_labelled: the number of walks changed between two runs");
		ap_free(S, B);
		cap = (n_lchunks > l_chunks ? n_lchunks : h) * AP_CH;          // (every walk has a start: the starts bound the records)
	}
	{
		uint64_t n_rec = 0;
		const int rc = ap_compact(S, v, B, h, h, &d_rec, &n_rec, nullptr);
		if (rc != SDT_OK) return rc;
		GCHK(hipStreamSynchronize(v.stream));
		ap_free(S, B);
		h = n_rec;
	}
	// components: removeSingleTips -- tip and end node of every walk; removeMinorTips -- the chains a walk can cross
	if (thin) {
		if (h) hipLaunchKernelGGL(k_uf_records, dim3(sdti::scan_grid(v.cu_count, h)), dim3(TPB), 0, v.stream, parent, d_rec, (uint64_t)h, 3, 1, 2, 0);
	} else {
		// every live port of every node that is neither linear nor deleted: listed, then walked one lane per port
		unsigned long long *d_list = nullptr, *d_lcur, n_lchunks = 0;
		GCHK(S.alloc(&d_lcur, 8));
		unsigned long long l_chunks = nn / (AP_CH - 64) + 1 + (unsigned long long)v.cu_count * 8 * (TPB / 64);
		for (int attempt = 0; attempt < 2; attempt++) {
			GCHK(S.alloc(&d_list, l_chunks * AP_CH * 8));
			GCHK(hipMemsetAsync(d_lcur, 0, 8, v.stream));
			LAUNCH_NW(v, k_port_starts, g, d_list, ApOut{d_lcur, l_chunks, nullptr, d_list});
			GCHK(hipGetLastError());
			GCHK(hipMemcpyAsync(&n_lchunks, d_lcur, 8, hipMemcpyDeviceToHost, v.stream));
			GCHK(hipStreamSynchronize(v.stream));
			if (n_lchunks <= l_chunks) break;
			if (attempt) return fail(SDT_ESTATE, "sdt_gpu_tip_walks_labelled: the number of ports changed between two runs");
			(void)hipFree(S.release(d_list));
			l_chunks = n_lchunks;
		}
		const unsigned long long n_list = n_lchunks * AP_CH;
		if (n_list) LAUNCH_NW(v, k_port_union_list, sdti::scan_grid(v.cu_count, n_list), *v.d_idx, v.K, cut_len, d_list, n_list, parent, v.d_stats);
		GCHK(hipGetLastError());
		GCHK(hipStreamSynchronize(v.stream));
		(void)hipFree(S.release(d_list)); (void)hipFree(S.release(d_lcur));
	}
	GCHK(hipGetLastError());
	int rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "sdt_gpu_tip_walks_labelled: %llu chains left the graph", (unsigned long long)v.h_stats->probe_fail);
	rc = label_sort_keep(c, v, parent, d_rec, h, h, 3);
	if (rc != SDT_OK) return rc;
	*n_records = h;
	return SDT_OK;
}

int sdt_gpu_minor_out_labelled(sdt_ctx *c, double threshold, uint64_t *n_junctions, uint64_t *n_records)
{
	if (!c || !n_junctions || !n_records) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	if (!*v.d_idx || *v.idx_slots != v.slots) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply (or sdt_gpu_set_node_index) first");
	const uint64_t nn = *v.idx_n;
	if (nn >= 0xFFFFFFF0ULL) return fail(SDT_EINVAL, "component labels are 32-bit node indices: %llu nodes", (unsigned long long)nn);
	HIPCHK(hipSetDevice(v.device));
	Scratch S;
	uint32_t *parent;
	uint8_t *d_need, *d_flag;
	unsigned long long *d_cur, h1 = 0, h2 = 0;
	GCHK(S.alloc(&parent, (nn + 1) * 4)); GCHK(S.alloc(&d_cur, 8));
	GCHK(S.alloc(&d_need, nn + 1)); GCHK(S.alloc(&d_flag, nn + 1));
	hipLaunchKernelGGL(k_uf_init, dim3(sdti::scan_grid(v.cu_count, nn + 1)), dim3(TPB), 0, v.stream, parent, nn + 1);
	const int g = sdti::scan_grid(v.cu_count, v.slots);
	// (one node in twenty has a record in the transcriptome jobs measured; the chunks add a third: room for one in sixteen, a second
	// attempt with what the first one counted otherwise)
	uint64_t cap = nn / 16 + 4096;
	uint64_t *d_rec = nullptr;
	ApBuf B;
	for (int attempt = 0; attempt < 2; attempt++) {
		int rc = ap_alloc(S, v, B, cap, 14);
		if (rc != SDT_OK) return rc;
		GCHK(hipMemsetAsync(d_need, 0, nn + 1, v.stream));
		GCHK(hipMemsetAsync(d_flag, 0, nn + 1, v.stream));
		LAUNCH_NW(v, k_minor_out_junctions, g, *v.d_idx, v.K, threshold, d_need, d_flag, B.chunks, 0ULL, (unsigned long long *)nullptr, v.d_stats, 14, ap_out(B));
		GCHK(hipGetLastError());
		GCHK(hipMemcpyAsync(&h1, B.cursor, 8, hipMemcpyDeviceToHost, v.stream));
		LAUNCH_NW(v, k_minor_out_candidates, g, *v.d_idx, v.K, d_need, d_flag, B.chunks, 0ULL, (unsigned long long *)nullptr, v.d_stats, 14, ap_out(B));
		GCHK(hipGetLastError());
		GCHK(hipMemcpyAsync(&h2, B.cursor, 8, hipMemcpyDeviceToHost, v.stream));
		rc = sdti::sync_stats(c);
		if (rc != SDT_OK) return fail(SDT_ESTATE, "sdt_gpu_minor_out_labelled: %llu links point at k-mers that are not nodes", (unsigned long long)v.h_stats->probe_fail);
		if (h2 <= B.cap_chunks) break;
		if (attempt) return fail(SDT_ESTATE, "sdt_gpu_minor_out_labelled: the number of records changed between two runs");
		ap_free(S, B);
		cap = h2 * AP_CH;
	}
	{
		// (h1, h2 are chunk counts so far: the junctions' chunks come first, packing keeps the chunk order)
		uint64_t n_rec = 0, n_junc = 0;
		const int rc = ap_compact(S, v, B, h2, h1, &d_rec, &n_rec, &n_junc);
		if (rc != SDT_OK) return rc;
		GCHK(hipStreamSynchronize(v.stream));
		ap_free(S, B);
		h1 = n_junc; h2 = n_rec;
	}
	// a visit reads and writes its junction, the junction's neighbours and the neighbours of those it may cut: unite every record's
	// node with its eight neighbours (junction records and the records of the neighbours to cut alike)
	if (h2) hipLaunchKernelGGL(k_uf_records, dim3(sdti::scan_grid(v.cu_count, h2)), dim3(TPB), 0, v.stream, parent, d_rec, (uint64_t)h2, 14, 1, 9, 1);
	GCHK(hipGetLastError());
	const int rc = label_sort_keep(c, v, parent, d_rec, h2, h1, 14);
	if (rc != SDT_OK) return rc;
	*n_junctions = h1;
	*n_records = h2;
	return SDT_OK;
}

// removeMinorOut's commit on the records sdt_gpu_minor_out_labelled left on the device (they stay there: sdt_gpu_fetch_records still
// works afterwards).  One lane walks a component, at about a microsecond per dependent access (most are first touches of a node:
// HBM latency); components of more than max_component visits are left alone -- their records wait for sdt_gpu_fetch_skipped, the
// host's threads are the better place for them.  Two halves, so that the host can work on those while the device walks the rest:
// _begin finds the components, gathers the records of the long ones and LAUNCHES the visits; _finish waits, re-marks and lists
// the written nodes.
static void mo_pending_free(sdti::GraphExt *gx)
{
	for (void **q : {(void **)&gx->mo_dirty, (void **)&gx->mo_cnt, (void **)&gx->mo_recidx, (void **)&gx->mo_cstart})
		if (*q) { (void)hipFree(*q); *q = nullptr; }
	gx->mo_pending = false;
}

int sdt_gpu_minor_out_commit_begin(sdt_ctx *c, double threshold, uint64_t max_component, uint64_t *largest, uint64_t *n_skipped, uint64_t *n_skipped_records)
{
	if (!c || !largest || !n_skipped || !n_skipped_records) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->d_slot_of || !*v.d_idx || *v.idx_slots != v.slots || *v.idx_n != gx->n_nodes) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply first");
	if (!gx->d_result || gx->result_stride != 14) return fail(SDT_ESTATE, "call sdt_gpu_minor_out_labelled first (its records must still be on the device)");
	const uint64_t nn = gx->n_nodes, nj = gx->result_labelled, nr = gx->result_words / 14;
	if (nn >= 0xFFFFFFF0ULL || nr >= 0xFFFFFFF0ULL) return fail(SDT_EINVAL, "commit on the device: 32-bit record and node indices");
	HIPCHK(hipSetDevice(v.device));
	*largest = *n_skipped = *n_skipped_records = 0;
	const bool timing = sdt_env("SDT_TIMING") != nullptr;
	struct timespec ts0_; clock_gettime(CLOCK_MONOTONIC, &ts0_);
	const double t_call = ts0_.tv_sec * 1e3 + ts0_.tv_nsec * 1e-6;
	auto tick = [&](const char *what) {              // (SDT_TIMING: where the call's time goes, on stderr; waits for the stream)
		if (!timing) return;
		(void)hipStreamSynchronize(v.stream);
		struct timespec t_;
		clock_gettime(CLOCK_MONOTONIC, &t_);
		fprintf(stderr, "[device]       commit: %s at %.1f ms\n", what, t_.tv_sec * 1e3 + t_.tv_nsec * 1e-6 - t_call);
	};
	if (gx->mo_pending) { (void)hipStreamSynchronize(v.stream); mo_pending_free(gx); }
	if (gx->d_wnode) { (void)hipFree(gx->d_wnode); gx->d_wnode = nullptr; }
	if (gx->d_wl) { (void)hipFree(gx->d_wl); gx->d_wl = nullptr; }
	if (gx->d_wr) { (void)hipFree(gx->d_wr); gx->d_wr = nullptr; }
	if (gx->d_skipped) { (void)hipFree(gx->d_skipped); gx->d_skipped = nullptr; }
	gx->n_written = gx->n_skipped = 0;
	Scratch S;
	uint32_t *flag, *rank, *cstart, *recidx;
	uint8_t *dirty;
	unsigned long long *d_cnt, h_largest = 0;                    // d_cnt: off, errors, marked, written, largest
	GCHK(S.alloc(&d_cnt, 5 * 8));
	GCHK(hipMemsetAsync(d_cnt, 0, 5 * 8, v.stream));
	GCHK(S.alloc(&recidx, (nn + 1) * 4)); GCHK(S.alloc(&dirty, nn + 1));
	GCHK(hipMemsetAsync(dirty, 0, nn + 1, v.stream));
	tick("buffers");
	if (!nj) {                                                   // nothing to visit: _finish reports zeros
		GCHK(S.alloc(&cstart, 8));
		gx->mo_dirty = (uint8_t *)S.release(dirty); gx->mo_cnt = (unsigned long long *)S.release(d_cnt);
		gx->mo_recidx = (uint32_t *)S.release(recidx); gx->mo_cstart = (uint32_t *)S.release(cstart);
		gx->mo_pending = true;
		return SDT_OK;
	}
	uint64_t ncomp = 0;
	{
	// (the temporaries of this block are let go BEFORE the visits are launched: freeing a block waits for the device, and at the
	// end of the call that would be a wait for the visits -- the host would start on the long components a quarter of a second late)
	Scratch T;
	GCHK(T.alloc(&flag, (nj + 1) * 4)); GCHK(T.alloc(&rank, (nj + 1) * 4));
	hipLaunchKernelGGL(k_mo_comp_flags, dim3(sdti::scan_grid(v.cu_count, nj + 1)), dim3(TPB), 0, v.stream, gx->d_result, nj, 14, flag);
	GCHK(hipGetLastError());
	int rc = exclusive_scan<uint32_t>(v, flag, rank, nj + 1);                   // rank[nj] = number of components
	if (rc != SDT_OK) return rc;
	uint32_t ncomp32 = 0;
	GCHK(hipMemcpyAsync(&ncomp32, rank + nj, 4, hipMemcpyDeviceToHost, v.stream));
	GCHK(hipStreamSynchronize(v.stream));
	ncomp = ncomp32;
	GCHK(S.alloc(&cstart, (ncomp + 1) * 4));
	hipLaunchKernelGGL(k_mo_comp_starts, dim3(sdti::scan_grid(v.cu_count, nj)), dim3(TPB), 0, v.stream, flag, rank, nj, cstart);
	GCHK(hipGetLastError());
	const uint32_t nj32 = (uint32_t)nj;
	GCHK(hipMemcpyAsync(cstart + ncomp, &nj32, 4, hipMemcpyHostToDevice, v.stream));
	hipLaunchKernelGGL(k_mo_comp_largest, dim3(sdti::scan_grid(v.cu_count, ncomp)), dim3(TPB), 0, v.stream, cstart, ncomp, d_cnt + 4);
	GCHK(hipGetLastError());
	GCHK(hipMemcpyAsync(&h_largest, d_cnt + 4, 8, hipMemcpyDeviceToHost, v.stream));
	GCHK(hipMemsetAsync(recidx, 0, (nn + 1) * 4, v.stream));
	hipLaunchKernelGGL(k_mo_recidx, dim3(sdti::scan_grid(v.cu_count, nr)), dim3(TPB), 0, v.stream, gx->d_result, nr, 14, recidx);
	GCHK(hipGetLastError());
	GCHK(hipStreamSynchronize(v.stream));
	*largest = h_largest;
	tick("components + record index");
	if (h_largest > max_component) {
		// the records of the components that are left alone: their junction records in order, then the records of the neighbours
		// they may cut (the host's commit finds the neighbours of a cut node there instead of looking them up)
		uint32_t *sel, *pos, *size_of, *sel2, *pos2, nsk = 0, nsk2 = 0;
		GCHK(T.alloc(&sel, (nj + 1) * 4)); GCHK(T.alloc(&pos, (nj + 1) * 4));
		hipLaunchKernelGGL(k_mo_skipped_sel, dim3(sdti::scan_grid(v.cu_count, nj + 1)), dim3(TPB), 0, v.stream, flag, rank, cstart, nj, max_component, sel);
		GCHK(hipGetLastError());
		rc = exclusive_scan<uint32_t>(v, sel, pos, nj + 1);
		if (rc != SDT_OK) return rc;
		GCHK(hipMemcpyAsync(&nsk, pos + nj, 4, hipMemcpyDeviceToHost, v.stream));
		const uint64_t nc = nr - nj;
		GCHK(T.alloc(&size_of, (nn + 1) * 4)); GCHK(T.alloc(&sel2, (nc + 1) * 4)); GCHK(T.alloc(&pos2, (nc + 1) * 4));
		GCHK(hipMemsetAsync(size_of, 0, (nn + 1) * 4, v.stream));
		hipLaunchKernelGGL(k_mo_label_sizes, dim3(sdti::scan_grid(v.cu_count, ncomp)), dim3(TPB), 0, v.stream, gx->d_result, 14, cstart, ncomp, size_of);
		GCHK(hipGetLastError());
		hipLaunchKernelGGL(k_mo_skipped_sel2, dim3(sdti::scan_grid(v.cu_count, nc + 1)), dim3(TPB), 0, v.stream, gx->d_result, 14, nj, nr, size_of, max_component, sel2);
		GCHK(hipGetLastError());
		rc = exclusive_scan<uint32_t>(v, sel2, pos2, nc + 1);
		if (rc != SDT_OK) return rc;
		GCHK(hipMemcpyAsync(&nsk2, pos2 + nc, 4, hipMemcpyDeviceToHost, v.stream));
		GCHK(hipStreamSynchronize(v.stream));
		uint64_t *sk;
		GCHK(S.alloc(&sk, ((uint64_t)nsk + nsk2 + 1) * 14 * 8));
		hipLaunchKernelGGL(k_mo_skipped_gather, dim3(sdti::scan_grid(v.cu_count, nj * 14)), dim3(TPB), 0, v.stream, gx->d_result, 14, sel, pos, nj, sk);
		GCHK(hipGetLastError());
		if (nc) hipLaunchKernelGGL(k_mo_skipped_gather, dim3(sdti::scan_grid(v.cu_count, nc * 14)), dim3(TPB), 0, v.stream, gx->d_result + nj * 14, 14, sel2, pos2, nc, sk + (uint64_t)nsk * 14);
		GCHK(hipGetLastError());
		GCHK(hipStreamSynchronize(v.stream));                // (sdt_gpu_fetch_skipped copies on the other stream)
		gx->d_skipped = (uint64_t *)S.release(sk);
		gx->n_skipped = (uint64_t)nsk + nsk2;
		*n_skipped = nsk;
		*n_skipped_records = (uint64_t)nsk + nsk2;
		tick("long components gathered");
	}
	}
	// the visits and the re-marking: launched, not waited for
	{
		const uint64_t blocks = (ncomp + TPB - 1) / TPB;
		const int g = (int)(blocks < (uint64_t)v.cu_count * 32 ? (blocks ? blocks : 1) : (uint64_t)v.cu_count * 32);
		LAUNCH_NW(v, k_mo_commit, g, gx->d_slot_of, v.K, threshold, gx->d_result, 14, cstart, ncomp, recidx, dirty, d_cnt, max_component);
		GCHK(hipGetLastError());
	}
	LAUNCH_NW(v, k_mo_mark, sdti::scan_grid(v.cu_count, nn), gx->d_slot_of, nn, dirty, d_cnt);
	GCHK(hipGetLastError());
	gx->mo_dirty = (uint8_t *)S.release(dirty); gx->mo_cnt = (unsigned long long *)S.release(d_cnt);
	gx->mo_recidx = (uint32_t *)S.release(recidx); gx->mo_cstart = (uint32_t *)S.release(cstart);
	gx->mo_pending = true;
	return SDT_OK;
}

int sdt_gpu_minor_out_commit_finish(sdt_ctx *c, uint64_t *off, uint64_t *linear, uint64_t *n_written)
{
	if (!c || !off || !linear || !n_written) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->mo_pending) return fail(SDT_ESTATE, "call sdt_gpu_minor_out_commit_begin first");
	HIPCHK(hipSetDevice(v.device));
	*off = *linear = *n_written = 0;
	const uint64_t nn = gx->n_nodes;
	unsigned long long h_cnt[5] = {0, 0, 0, 0, 0};
	GCHK(hipMemcpyAsync(h_cnt, gx->mo_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, v.stream));
	GCHK(hipStreamSynchronize(v.stream));
	if (h_cnt[1]) { mo_pending_free(gx); return fail(SDT_ESTATE, "sdt_gpu_minor_out_commit: %llu cuts found no record of the node they cut", h_cnt[1]); }
	const uint64_t nw = h_cnt[3];
	Scratch S;
	unsigned long long *d_cur;
	GCHK(S.alloc(&d_cur, 8));
	GCHK(hipMemsetAsync(d_cur, 0, 8, v.stream));
	uint64_t *wn; uint32_t *wl, *wr;
	GCHK(S.alloc(&wn, (nw + 1) * 8)); GCHK(S.alloc(&wl, (nw + 1) * 4)); GCHK(S.alloc(&wr, (nw + 1) * 4));
	if (nw) {
		LAUNCH_NW(v, k_mo_emit, sdti::scan_grid(v.cu_count, nn), gx->d_slot_of, nn, gx->mo_dirty, d_cur, nw, wn, wl, wr);
		GCHK(hipGetLastError());
	}
	GCHK(hipStreamSynchronize(v.stream));
	mo_pending_free(gx);
	gx->d_wnode = (uint64_t *)S.release(wn); gx->d_wl = (uint32_t *)S.release(wl); gx->d_wr = (uint32_t *)S.release(wr);
	gx->n_written = nw;
	*off = h_cnt[0];
	*linear = h_cnt[2];
	*n_written = nw;
	return SDT_OK;
}

int sdt_gpu_minor_out_commit(sdt_ctx *c, double threshold, uint64_t max_component, uint64_t *largest, uint64_t *off, uint64_t *linear, uint64_t *n_written,
                             uint64_t *n_skipped, uint64_t *n_skipped_records)
{
	if (!off || !linear || !n_written) return fail(SDT_EINVAL, "NULL argument");
	const int rc = sdt_gpu_minor_out_commit_begin(c, threshold, max_component, largest, n_skipped, n_skipped_records);
	return rc != SDT_OK ? rc : sdt_gpu_minor_out_commit_finish(c, off, linear, n_written);
}

int sdt_gpu_fetch_skipped(sdt_ctx *c, uint64_t *dst, uint64_t n_records)
{
	if (!c || (n_records && !dst)) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (n_records != gx->n_skipped) return fail(SDT_EINVAL, "the last commit left %llu records to the host, asked for %llu", (unsigned long long)gx->n_skipped, (unsigned long long)n_records);
	HIPCHK(hipSetDevice(v.device));
	int rc = SDT_OK;
	if (n_records) rc = sdti::d2h_big(v.copy_stream, dst, gx->d_skipped, n_records * 14 * 8);
	// (no hipFree here: freeing waits for the device, and the visits of _begin may be running -- _begin / the context let go of it)
	return rc;
}

int sdt_gpu_fetch_written(sdt_ctx *c, uint64_t *node, uint32_t *l_links, uint32_t *r_flags, uint64_t n)
{
	if (!c || (n && (!node || !l_links || !r_flags))) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (n != gx->n_written) return fail(SDT_EINVAL, "the last commit wrote %llu nodes, asked for %llu", (unsigned long long)gx->n_written, (unsigned long long)n);
	HIPCHK(hipSetDevice(v.device));
	int rc = SDT_OK;
	if (n) rc = sdti::d2h_big(v.copy_stream, node, gx->d_wnode, n * 8);
	if (n && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, l_links, gx->d_wl, n * 4);
	if (n && rc == SDT_OK) rc = sdti::d2h_big(v.copy_stream, r_flags, gx->d_wr, n * 4);
	if (gx->d_wnode) (void)hipFree(gx->d_wnode);
	if (gx->d_wl) (void)hipFree(gx->d_wl);
	if (gx->d_wr) (void)hipFree(gx->d_wr);
	gx->d_wnode = nullptr; gx->d_wl = gx->d_wr = nullptr;
	gx->n_written = 0;
	return rc;
}

// ---- kmer2edges on the device (node2edge.c:46-561) ----------------------------------------------------------------------
int sdt_gpu_build_edges(sdt_ctx *c, uint64_t *n_edges, uint64_t *num_ed, uint64_t *n_bases)
{
	if (!c || !n_edges || !num_ed || !n_bases) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (!gx->d_slot_of || !*v.d_idx || *v.idx_slots != v.slots || *v.idx_n != gx->n_nodes) return fail(SDT_ESTATE, "call sdt_gpu_layout_apply first");
	const uint64_t n = gx->n_nodes;
	if (n >= 0xFFFFFFF0ULL) return fail(SDT_EINVAL, "edge building on the device: 32-bit node indices, %llu nodes", (unsigned long long)n);
	HIPCHK(hipSetDevice(v.device));
	if (gx->d_result) { (void)hipFree(gx->d_result); gx->d_result = nullptr; gx->result_words = 0; }
	if (gx->d_seq) { (void)hipFree(gx->d_seq); gx->d_seq = nullptr; gx->seq_bytes = 0; }
	if (gx->d_pw) { (void)hipFree(gx->d_pw); gx->d_pw = nullptr; gx->pw_n = 0; }
	Scratch S;
	const uint64_t m = n ? n : 1;
	uint32_t *flag, *srank, *start_node;
	uint64_t *pw;
	unsigned int *d_asym;
	GCHK(S.alloc(&flag, (m + 1) * 4)); GCHK(S.alloc(&srank, (m + 1) * 4)); GCHK(S.alloc(&pw, m * 8)); GCHK(S.alloc(&d_asym, 4));
	GCHK(hipMemsetAsync(flag, 0, (m + 1) * 4, v.stream));
	GCHK(hipMemsetAsync(d_asym, 0, 4, v.stream));
	LAUNCH_NW(v, k_edge_starts, sdti::scan_grid(v.cu_count, m), gx->d_slot_of, n, flag, pw);
	GCHK(hipGetLastError());
	int rc = exclusive_scan<uint32_t>(v, flag, srank, n + 1);                  // srank[n] = number of start nodes
	if (rc != SDT_OK) return rc;
	uint32_t nstarts = 0;
	GCHK(hipMemcpy(&nstarts, srank + n, 4, hipMemcpyDeviceToHost));
	const uint64_t nports = (uint64_t)nstarts * 8;
	PortRec *ports;
	uint32_t *w_edge, *w_id, *e_scan, *id_scan;
	uint64_t *w_len, *len_scan;
	GCHK(S.alloc(&start_node, ((uint64_t)nstarts + 1) * 4)); GCHK(S.alloc(&ports, (nports + 1) * sizeof(PortRec)));
	GCHK(S.alloc(&w_edge, (nports + 1) * 4)); GCHK(S.alloc(&w_id, (nports + 1) * 4)); GCHK(S.alloc(&w_len, (nports + 1) * 8));
	GCHK(S.alloc(&e_scan, (nports + 1) * 4)); GCHK(S.alloc(&id_scan, (nports + 1) * 4)); GCHK(S.alloc(&len_scan, (nports + 1) * 8));
	GCHK(hipMemsetAsync(w_edge + nports, 0, 4, v.stream)); GCHK(hipMemsetAsync(w_id + nports, 0, 4, v.stream)); GCHK(hipMemsetAsync(w_len + nports, 0, 8, v.stream));
	hipLaunchKernelGGL(k_edge_start_nodes, dim3(sdti::scan_grid(v.cu_count, m)), dim3(TPB), 0, v.stream, flag, srank, n, start_node);
	if (nports) LAUNCH_NW(v, k_edge_ports_ordered, sdti::scan_grid(v.cu_count, nports), *v.d_idx, gx->d_slot_of, start_node, nports, v.K, n + 1, ports, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "sdt_gpu_build_edges: %llu chains leave the graph or never end", (unsigned long long)v.h_stats->probe_fail);
	if (nports) hipLaunchKernelGGL(k_edge_emit, dim3(sdti::scan_grid(v.cu_count, nports)), dim3(TPB), 0, v.stream, ports, start_node, flag, srank, nports, w_edge, w_id, w_len, d_asym);
	GCHK(hipGetLastError());
	unsigned int asym = 0;
	GCHK(hipMemcpyAsync(&asym, d_asym, 4, hipMemcpyDeviceToHost, v.stream));
	GCHK(hipStreamSynchronize(v.stream));
	if (asym) return fail(SDT_ESTATE, "sdt_gpu_build_edges: a chain does not lead back to the port it was entered from (build the edges sequentially)");
	rc = exclusive_scan<uint32_t>(v, w_edge, e_scan, nports + 1);
	if (rc == SDT_OK) rc = exclusive_scan<uint32_t>(v, w_id, id_scan, nports + 1);
	if (rc == SDT_OK) rc = exclusive_scan<uint64_t>(v, w_len, len_scan, nports + 1);
	if (rc != SDT_OK) return rc;
	uint32_t ne = 0, ids = 0;
	uint64_t nb = 0;
	GCHK(hipMemcpy(&ne, e_scan + nports, 4, hipMemcpyDeviceToHost));
	GCHK(hipMemcpy(&ids, id_scan + nports, 4, hipMemcpyDeviceToHost));
	GCHK(hipMemcpy(&nb, len_scan + nports, 8, hipMemcpyDeviceToHost));
	const int RW = 4 + 2 * v.nw;
	uint64_t *erec;
	unsigned char *seq;
	GCHK(S.alloc(&erec, ((uint64_t)ne + 1) * RW * 8)); GCHK(S.alloc(&seq, nb + 16));
	if (nports) LAUNCH_NW(v, k_edge_stamp, sdti::scan_grid(v.cu_count, nports), *v.d_idx, gx->d_slot_of, v.K, ports, start_node, nports, w_edge, e_scan, id_scan, len_scan, pw, seq, erec, v.d_stats);
	GCHK(hipGetLastError());
	rc = sdti::sync_stats(c);
	if (rc != SDT_OK) return fail(SDT_ESTATE, "sdt_gpu_build_edges: %llu chains changed between the two walks", (unsigned long long)v.h_stats->probe_fail);
	gx->d_result = (uint64_t *)S.release(erec);
	gx->result_words = (uint64_t)ne * RW;
	gx->d_seq = (unsigned char *)S.release(seq);
	gx->seq_bytes = nb;
	gx->d_pw = (uint64_t *)S.release(pw);
	gx->pw_n = n;
	*n_edges = ne;
	*num_ed = ids;
	*n_bases = nb;
	return SDT_OK;
}

int sdt_gpu_fetch_edge_bases(sdt_ctx *c, char *dst, uint64_t nbytes)
{
	if (!c || (nbytes && !dst)) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (nbytes != gx->seq_bytes) return fail(SDT_EINVAL, "the edges have %llu bases, asked for %llu", (unsigned long long)gx->seq_bytes, (unsigned long long)nbytes);
	HIPCHK(hipSetDevice(v.device));
	int rc = SDT_OK;
	if (nbytes) rc = sdti::d2h_big(v.copy_stream, dst, gx->d_seq, nbytes);
	if (gx->d_seq) (void)hipFree(gx->d_seq);
	gx->d_seq = nullptr;
	gx->seq_bytes = 0;
	return rc;
}

int sdt_gpu_fetch_records(sdt_ctx *c, uint64_t *dst, uint64_t nwords)
{
	if (!c || (nwords && !dst)) return fail(SDT_EINVAL, "NULL argument");
	const GraphView v = sdti::graph_view(c);
	sdti::GraphExt *gx = ext_of(v);
	if (nwords != gx->result_words) return fail(SDT_EINVAL, "the last dry run left %llu words, asked for %llu", (unsigned long long)gx->result_words, (unsigned long long)nwords);
	HIPCHK(hipSetDevice(v.device));
	int rc = SDT_OK;
	if (nwords) rc = sdti::d2h_big(v.copy_stream, dst, gx->d_result, nwords * 8);
	if (gx->d_result) (void)hipFree(gx->d_result);
	gx->d_result = nullptr;
	gx->result_words = 0;
	return rc;
}

// ---- graph-cleaning dry runs on the device mirror of the host graph ---------------------------------------
static int upload_keys(sdt_ctx *c, const uint64_t *keys, uint64_t n, uint64_t **d_k)
{
	const GraphView v = sdti::graph_view(c);
	HIPCHK(hipMalloc((void **)d_k, (n ? n : 1) * v.nw * sizeof(uint64_t)));
	const int rc = sdti::h2d_big(v.copy_stream, *d_k, keys, n * v.nw * sizeof(uint64_t));
	if (rc != SDT_OK) { (void)hipFree(*d_k); *d_k = nullptr; }
	return rc;
}

int sdt_gpu_set_node_index(sdt_ctx *c, const uint64_t *keys, uint64_t n)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || (n && !keys))
		return fail(SDT_EINVAL, "NULL argument");
	HIPCHK(hipSetDevice(v.device));
	HIPCHK(hipStreamSynchronize(v.stream));
	if ((*v.d_idx)) HIPCHK(hipFree((*v.d_idx)));
	(*v.d_idx) = nullptr;
	(*v.idx_slots) = (*v.idx_n) = 0;
	HIPCHK(hipMalloc((void **)&(*v.d_idx), v.slots * sizeof(uint64_t)));
	HIPCHK(hipMemsetAsync((*v.d_idx), 0xFF, v.slots * sizeof(uint64_t), v.stream));
	uint64_t *d_k = nullptr;
	int rc = upload_keys(c, keys, n, &d_k);
	if (rc != SDT_OK) return rc;
	const int g = sdti::scan_grid(v.cu_count, n ? n : 1);
	if (v.nw == 1) hipLaunchKernelGGL(k_set_index<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), d_k, n, (*v.d_idx), v.d_stats);
	else if (v.nw == 2) hipLaunchKernelGGL(k_set_index<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), d_k, n, (*v.d_idx), v.d_stats);
	else hipLaunchKernelGGL(k_set_index<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), d_k, n, (*v.d_idx), v.d_stats);
	hipError_t le = hipGetLastError();
	rc = le == hipSuccess ? sdti::sync_stats(c) : fail(SDT_EHIP, "k_set_index: %s", hipGetErrorString(le));
	(void)hipFree(d_k);
	if (rc != SDT_OK)
		return fail(SDT_ESTATE, "sdt_gpu_set_node_index: %llu nodes are not in the table", (unsigned long long)v.h_stats->probe_fail);
	(*v.idx_slots) = v.slots;
	(*v.idx_n) = n;
	return SDT_OK;
}

int sdt_gpu_update_nodes(sdt_ctx *c, const uint64_t *keys, const uint32_t *l_links, const uint32_t *r_flags, uint64_t n)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || (n && (!keys || !l_links || !r_flags)))
		return fail(SDT_EINVAL, "NULL argument");
	if (!n)
		return SDT_OK;
	HIPCHK(hipSetDevice(v.device));
	uint64_t *d_k = nullptr;
	uint32_t *d_l = nullptr, *d_r = nullptr;
	int rc = upload_keys(c, keys, n, &d_k);
	if (rc != SDT_OK) return rc;
	hipError_t e = hipMalloc((void **)&d_l, n * 4);
	if (e == hipSuccess) e = hipMalloc((void **)&d_r, n * 4);
	if (e == hipSuccess) e = hipMemcpyAsync(d_l, l_links, n * 4, hipMemcpyHostToDevice, v.stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_r, r_flags, n * 4, hipMemcpyHostToDevice, v.stream);
	if (e == hipSuccess) {
		const int g = sdti::scan_grid(v.cu_count, n);
		if (v.nw == 1) hipLaunchKernelGGL(k_update_nodes<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), d_k, d_l, d_r, n, v.d_stats);
		else if (v.nw == 2) hipLaunchKernelGGL(k_update_nodes<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), d_k, d_l, d_r, n, v.d_stats);
		else hipLaunchKernelGGL(k_update_nodes<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), d_k, d_l, d_r, n, v.d_stats);
		e = hipGetLastError();
	}
	rc = e == hipSuccess ? sdti::sync_stats(c) : fail(SDT_EHIP, "sdt_gpu_update_nodes: %s", hipGetErrorString(e));
	(void)hipFree(d_k);
	if (d_l) (void)hipFree(d_l);
	if (d_r) (void)hipFree(d_r);
	if (rc != SDT_OK && e == hipSuccess)
		return fail(SDT_ESTATE, "sdt_gpu_update_nodes: %llu nodes are not in the table", (unsigned long long)v.h_stats->probe_fail);
	return rc;
}

int sdt_gpu_tip_walks(sdt_ctx *c, int thin, int cut_len, uint64_t *end_idx, uint8_t *info, uint64_t n)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || !end_idx || !info)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(*v.d_idx) || (*v.idx_slots) != v.slots)
		return fail(SDT_ESTATE, "call sdt_gpu_set_node_index first");
	if (n != (*v.idx_n))
		return fail(SDT_EINVAL, "the node index holds %llu nodes, the output arrays %llu", (unsigned long long)(*v.idx_n), (unsigned long long)n);
	HIPCHK(hipSetDevice(v.device));
	uint64_t *d_e = nullptr;
	uint8_t *d_i = nullptr;
	HIPCHK(hipMalloc((void **)&d_e, (n ? n : 1) * sizeof(uint64_t)));
	hipError_t e = hipMalloc((void **)&d_i, n ? n : 1);
	if (e == hipSuccess) {
		const int g = sdti::scan_grid(v.cu_count, v.slots);
		if (v.nw == 1) hipLaunchKernelGGL(k_tip_walks<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), v.K, thin, cut_len, d_e, d_i, v.d_stats, (uint64_t *)nullptr, 0ULL, (unsigned long long *)nullptr, 2);
		else if (v.nw == 2) hipLaunchKernelGGL(k_tip_walks<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), v.K, thin, cut_len, d_e, d_i, v.d_stats, (uint64_t *)nullptr, 0ULL, (unsigned long long *)nullptr, 2);
		else hipLaunchKernelGGL(k_tip_walks<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), v.K, thin, cut_len, d_e, d_i, v.d_stats, (uint64_t *)nullptr, 0ULL, (unsigned long long *)nullptr, 2);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(end_idx, d_e, n * sizeof(uint64_t), hipMemcpyDeviceToHost, v.stream);
	if (e == hipSuccess) e = hipMemcpyAsync(info, d_i, n, hipMemcpyDeviceToHost, v.stream);
	int rc = e == hipSuccess ? sdti::sync_stats(c) : fail(SDT_EHIP, "sdt_gpu_tip_walks: %s", hipGetErrorString(e));
	(void)hipFree(d_e);
	if (d_i) (void)hipFree(d_i);
	if (rc != SDT_OK && e == hipSuccess)
		return fail(SDT_ESTATE, "sdt_gpu_tip_walks: %llu walks left the graph (a link points at a k-mer that is not a node)",
		            (unsigned long long)v.h_stats->probe_fail);
	return rc;
}


int sdt_gpu_minor_out_dry(sdt_ctx *c, double threshold, uint64_t *records, uint64_t max_records, uint64_t *n_junctions, uint64_t *n_records)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || (!records && max_records) || !n_junctions || !n_records)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(*v.d_idx) || (*v.idx_slots) != v.slots)
		return fail(SDT_ESTATE, "call sdt_gpu_set_node_index first");
	HIPCHK(hipSetDevice(v.device));
	uint8_t *d_need = nullptr, *d_flag = nullptr;
	uint64_t *d_rec = nullptr;
	unsigned long long *d_cur = nullptr;
	unsigned long long h1 = 0, h2 = 0;
	int ret = SDT_OK;
	const uint64_t n = (*v.idx_n) ? (*v.idx_n) : 1, m = max_records ? max_records : 1;
#define MO_CHK(expr) do { hipError_t e5_ = (expr); if (e5_ != hipSuccess) { ret = fail(e5_ == hipErrorOutOfMemory ? SDT_ENOMEM : SDT_EHIP, "%s: %s", #expr, hipGetErrorString(e5_)); goto done; } } while (0)
	MO_CHK(hipMalloc((void **)&d_need, n));
	MO_CHK(hipMalloc((void **)&d_flag, n));
	MO_CHK(hipMalloc((void **)&d_rec, m * 9 * sizeof(uint64_t)));
	MO_CHK(hipMalloc((void **)&d_cur, sizeof(unsigned long long)));
	MO_CHK(hipMemsetAsync(d_need, 0, n, v.stream));
	MO_CHK(hipMemsetAsync(d_flag, 0, n, v.stream));
	MO_CHK(hipMemsetAsync(d_cur, 0, sizeof(unsigned long long), v.stream));
	{
		const int g = sdti::scan_grid(v.cu_count, v.slots);
		if (v.nw == 1) hipLaunchKernelGGL(k_minor_out_junctions<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), v.K, threshold, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		else if (v.nw == 2) hipLaunchKernelGGL(k_minor_out_junctions<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), v.K, threshold, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		else hipLaunchKernelGGL(k_minor_out_junctions<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), v.K, threshold, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		MO_CHK(hipGetLastError());
		MO_CHK(hipMemcpyAsync(&h1, d_cur, sizeof h1, hipMemcpyDeviceToHost, v.stream));
		if (v.nw == 1) hipLaunchKernelGGL(k_minor_out_candidates<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), v.K, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		else if (v.nw == 2) hipLaunchKernelGGL(k_minor_out_candidates<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), v.K, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		else hipLaunchKernelGGL(k_minor_out_candidates<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), v.K, d_need, d_flag, d_rec, (unsigned long long)max_records, d_cur, v.d_stats, 9);
		MO_CHK(hipGetLastError());
		MO_CHK(hipMemcpyAsync(&h2, d_cur, sizeof h2, hipMemcpyDeviceToHost, v.stream));
	}
	ret = sdti::sync_stats(c);
	if (ret != SDT_OK) {
		ret = fail(SDT_ESTATE, "sdt_gpu_minor_out_dry: %llu links point at k-mers that are not nodes", (unsigned long long)v.h_stats->probe_fail);
		goto done;
	}
	*n_junctions = h1;
	*n_records = h2;
	if (h2 > max_records) {
		ret = fail(SDT_EFULL, "record array holds %llu, the pass needs %llu", (unsigned long long)max_records, h2);
		goto done;
	}
	if (h2) MO_CHK(hipMemcpy(records, d_rec, h2 * 9 * sizeof(uint64_t), hipMemcpyDeviceToHost));
done:
#undef MO_CHK
	if (d_need) (void)hipFree(d_need);
	if (d_flag) (void)hipFree(d_flag);
	if (d_rec) (void)hipFree(d_rec);
	if (d_cur) (void)hipFree(d_cur);
	return ret;
}

int sdt_gpu_build_host_index(sdt_ctx *c, uint32_t *index, uint64_t index_slots)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || !index)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(*v.d_idx) || (*v.idx_slots) != v.slots)
		return fail(SDT_ESTATE, "call sdt_gpu_set_node_index first");
	if (index_slots < 2 * (*v.idx_n) || (index_slots & (index_slots - 1)))
		return fail(SDT_EINVAL, "index_slots must be a power of two >= 2 x nodes");
	if ((*v.idx_n) >= 0xFFFFFFFEULL)
		return fail(SDT_EINVAL, "%llu nodes do not fit 32-bit index entries", (unsigned long long)(*v.idx_n));
	HIPCHK(hipSetDevice(v.device));
	unsigned int *d_index = nullptr;
	HIPCHK(hipMalloc((void **)&d_index, index_slots * sizeof(unsigned int)));
	hipError_t e = hipMemsetAsync(d_index, 0, index_slots * sizeof(unsigned int), v.stream);
	if (e == hipSuccess) {
		const int g = sdti::scan_grid(v.cu_count, v.slots);
		if (v.nw == 1) hipLaunchKernelGGL(k_build_host_index<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), d_index, index_slots - 1);
		else if (v.nw == 2) hipLaunchKernelGGL(k_build_host_index<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), d_index, index_slots - 1);
		else hipLaunchKernelGGL(k_build_host_index<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), d_index, index_slots - 1);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(v.stream);
	int rc = SDT_OK;
	if (e == hipSuccess) rc = sdti::d2h_big(v.copy_stream, index, d_index, index_slots * sizeof(unsigned int));      // (pageable destination of gigabytes: staged copies)
	(void)hipFree(d_index);
	if (e != hipSuccess)
		return fail(SDT_EHIP, "sdt_gpu_build_host_index: %s", hipGetErrorString(e));
	return rc;
}

int sdt_gpu_edge_ports(sdt_ctx *c, uint64_t *records, uint64_t max_records, uint64_t *n_records)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || (!records && max_records) || !n_records)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(*v.d_idx) || (*v.idx_slots) != v.slots)
		return fail(SDT_ESTATE, "call sdt_gpu_set_node_index first");
	HIPCHK(hipSetDevice(v.device));
	uint64_t *d_rec = nullptr;
	unsigned long long *d_cur = nullptr, h = 0;
	const uint64_t m = max_records ? max_records : 1;
	HIPCHK(hipMalloc((void **)&d_rec, m * 17 * sizeof(uint64_t)));
	hipError_t e = hipMalloc((void **)&d_cur, sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, sizeof(unsigned long long), v.stream);
	if (e == hipSuccess) {
		const int g = sdti::scan_grid(v.cu_count, v.slots);
		if (v.nw == 1) hipLaunchKernelGGL(k_edge_ports<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), v.K, (*v.idx_n) + 1, d_rec, (unsigned long long)max_records, d_cur, v.d_stats);
		else if (v.nw == 2) hipLaunchKernelGGL(k_edge_ports<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), v.K, (*v.idx_n) + 1, d_rec, (unsigned long long)max_records, d_cur, v.d_stats);
		else hipLaunchKernelGGL(k_edge_ports<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), v.K, (*v.idx_n) + 1, d_rec, (unsigned long long)max_records, d_cur, v.d_stats);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(&h, d_cur, sizeof h, hipMemcpyDeviceToHost, v.stream);
	int rc = e == hipSuccess ? sdti::sync_stats(c) : fail(SDT_EHIP, "sdt_gpu_edge_ports: %s", hipGetErrorString(e));
	if (rc != SDT_OK && e == hipSuccess)
		rc = fail(SDT_ESTATE, "sdt_gpu_edge_ports: %llu chains leave the graph or never end", (unsigned long long)v.h_stats->probe_fail);
	if (rc == SDT_OK) {
		*n_records = h;
		if (h > max_records) rc = fail(SDT_EFULL, "record array holds %llu, the graph has %llu non-linear nodes", (unsigned long long)max_records, h);
		else if (h && hipMemcpy(records, d_rec, h * 17 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SDT_EHIP, "edge port records: copy failed");
	}
	(void)hipFree(d_rec);
	if (d_cur) (void)hipFree(d_cur);
	return rc;
}

int sdt_gpu_tip_walks_compact(sdt_ctx *c, int thin, int cut_len, uint64_t *records, uint64_t max_records, uint64_t *n_records)
{
	if (!c) return fail(SDT_EINVAL, "ctx is NULL");
	const GraphView v = sdti::graph_view(c);
	if (!c || (!records && max_records) || !n_records)
		return fail(SDT_EINVAL, "NULL argument");
	if (!(*v.d_idx) || (*v.idx_slots) != v.slots)
		return fail(SDT_ESTATE, "call sdt_gpu_set_node_index first");
	if ((*v.idx_n) >= (1ULL << 56))
		return fail(SDT_EINVAL, "node indices do not fit 56 bits");
	HIPCHK(hipSetDevice(v.device));
	uint64_t *d_rec = nullptr;
	unsigned long long *d_cur = nullptr, h = 0;
	const uint64_t m = max_records ? max_records : 1;
	HIPCHK(hipMalloc((void **)&d_rec, m * 2 * sizeof(uint64_t)));
	hipError_t e = hipMalloc((void **)&d_cur, sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemsetAsync(d_cur, 0, sizeof(unsigned long long), v.stream);
	if (e == hipSuccess) {
		const int g = sdti::scan_grid(v.cu_count, v.slots);
		if (v.nw == 1) hipLaunchKernelGGL(k_tip_walks<1>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<1>(v), (*v.d_idx), v.K, thin, cut_len, (uint64_t *)nullptr, (uint8_t *)nullptr, v.d_stats, d_rec, (unsigned long long)max_records, d_cur, 2);
		else if (v.nw == 2) hipLaunchKernelGGL(k_tip_walks<2>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<2>(v), (*v.d_idx), v.K, thin, cut_len, (uint64_t *)nullptr, (uint8_t *)nullptr, v.d_stats, d_rec, (unsigned long long)max_records, d_cur, 2);
		else hipLaunchKernelGGL(k_tip_walks<4>, dim3(g), dim3(TPB), 0, v.stream, sdti::table_of<4>(v), (*v.d_idx), v.K, thin, cut_len, (uint64_t *)nullptr, (uint8_t *)nullptr, v.d_stats, d_rec, (unsigned long long)max_records, d_cur, 2);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(&h, d_cur, sizeof h, hipMemcpyDeviceToHost, v.stream);
	int rc = e == hipSuccess ? sdti::sync_stats(c) : fail(SDT_EHIP, "sdt_gpu_tip_walks_compact: %s", hipGetErrorString(e));
	if (rc != SDT_OK && e == hipSuccess)
		rc = fail(SDT_ESTATE, "sdt_gpu_tip_walks_compact: %llu walks left the graph", (unsigned long long)v.h_stats->probe_fail);
	if (rc == SDT_OK) {
		*n_records = h;
		if (h > max_records) rc = fail(SDT_EFULL, "record array holds %llu, %llu nodes have a walk", (unsigned long long)max_records, h);
		else if (h && hipMemcpy(records, d_rec, h * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SDT_EHIP, "tip walk records: copy failed");
	}
	(void)hipFree(d_rec);
	if (d_cur) (void)hipFree(d_cur);
	return rc;
}


}  // extern "C"
