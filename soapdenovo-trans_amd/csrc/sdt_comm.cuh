// sdt_comm.cuh -- the exchange step of pass 1 across GPUs (host side), included by sdt_gpu.hip.
//
// The reference routes every chopped record to thread `hash_kmer % thrd_num` by letting each thread scan the whole
// batch (prlHashReads.c:77-90).  Across GPUs the unit that travels is the super-k-mer record (sdt_superkmer.cuh) and
// the owner of a k-mer is the rank that owns its level-1 bucket: ranks own contiguous ranges of the 256 buckets.
// Two transports behind one interface:
//   RCCL   grouped ncclSend / ncclRecv (size_t byte counts) on a stream of their own: xGMI inside a node.  librccl.so.1
//          is loaded lazily with dlopen (the same library a PyTorch process already has; none is needed for one rank).
//   SHM    POSIX shared memory + host staging.  For validation where several ranks share ONE GPU (RCCL refuses two ranks
//          per device): every line of the protocol above the byte mover is the same.
// Control data (chunk counts per bucket, counters, the kmerFreq bins) goes through allgather_host / allreduce_host.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include <atomic>
#include <utility>
#include <vector>

namespace sdt {

struct NcclId { char internal[128]; };               // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *NcclComm;

struct RcclApi {
	void *lib = nullptr;
	int (*GetUniqueId)(NcclId *) = nullptr;
	int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
	int (*CommDestroy)(NcclComm) = nullptr;
	int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
	int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_UINT8 = 1, NCCL_INT64 = 4, NCCL_SUM = 0;

inline RcclApi g_rccl;               // (one instance for all translation units of the library)

inline int rccl_load()
{
	if (g_rccl.lib)
		return SDT_OK;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!h)
		return fail(SDT_ENODEV, "cannot load librccl.so.1: %s", dlerror());
#define SDT_SYM(field, name)                                                         \
	do {                                                                             \
		*(void **)(&g_rccl.field) = dlsym(h, name);                                  \
		if (!g_rccl.field) return fail(SDT_ENODEV, "librccl: symbol %s missing", name); \
	} while (0)
	SDT_SYM(GetUniqueId, "ncclGetUniqueId");
	SDT_SYM(CommInitRank, "ncclCommInitRank");
	SDT_SYM(CommDestroy, "ncclCommDestroy");
	SDT_SYM(Send, "ncclSend");
	SDT_SYM(Recv, "ncclRecv");
	SDT_SYM(AllGather, "ncclAllGather");
	SDT_SYM(AllReduce, "ncclAllReduce");
	SDT_SYM(GroupStart, "ncclGroupStart");
	SDT_SYM(GroupEnd, "ncclGroupEnd");
	SDT_SYM(GetErrorString, "ncclGetErrorString");
#undef SDT_SYM
	g_rccl.lib = h;
	return SDT_OK;
}

#define NCCLCHK(expr)                                                                                  \
	do {                                                                                               \
		const int r_ = (expr);                                                                         \
		if (r_ != 0)                                                                                   \
			return fail(SDT_EHIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
	} while (0)

// ---- shared-memory transport ------------------------------------------------------------------------------------
struct ShmHeader {
	std::atomic<uint32_t> ready, arrived, generation, failed;
	uint32_t nranks;
	uint64_t ctrl_bytes, outbox_bytes;
};
constexpr size_t SHM_HEADER_BYTES = 4096;
constexpr size_t SHM_CTRL_BYTES = 64 * 1024;         // per rank: control messages (allgather / allreduce payloads)
constexpr double SHM_TIMEOUT_S = 300.0;

inline double comm_now()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

struct Comm {
	int kind = 0;                                    // 0 none, 1 RCCL, 2 shared memory
	int rank = 0, nranks = 1;
	hipStream_t xstream = nullptr;                   // the exchange runs here, beside the kernels
	NcclComm nccl = nullptr;
	void *d_ctrl = nullptr;                          // RCCL: device staging of control messages (nranks + 1 slots)
	void *h_ctrl = nullptr;                          // pinned
	// shm
	char shm_name[200] = "";
	uint8_t *shm = nullptr;
	size_t shm_bytes = 0, outbox_bytes = 0;
	// accounting (sdt_gpu_comm_stats)
	uint64_t bytes_sent = 0, bytes_recv = 0, exchanges = 0;
	double exchange_ms = 0;
	// RCCL: one event pair per exchange() call; harvest_time() adds the pairs that have completed, each exactly once
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
	size_t ev_used = 0, ev_harvested = 0;

	ShmHeader *hdr() const { return (ShmHeader *)shm; }
	uint8_t *ctrl(int r) const { return shm + SHM_HEADER_BYTES + (size_t)r * SHM_CTRL_BYTES; }
	uint8_t *outbox(int r) const { return shm + SHM_HEADER_BYTES + (size_t)nranks * SHM_CTRL_BYTES + (size_t)r * outbox_bytes; }

	int shm_barrier()
	{
		ShmHeader *h = hdr();
		const uint32_t gen = h->generation.load(std::memory_order_acquire);
		if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
			h->arrived.store(0, std::memory_order_relaxed);
			h->generation.fetch_add(1, std::memory_order_acq_rel);
			return SDT_OK;
		}
		const double t0 = comm_now();
		while (h->generation.load(std::memory_order_acquire) == gen) {
			if (h->failed.load(std::memory_order_relaxed))
				return fail(SDT_EHIP, "shared-memory transport: another rank failed");
			if (comm_now() - t0 > SHM_TIMEOUT_S) {
				h->failed.store(1);
				return fail(SDT_EHIP, "shared-memory transport: barrier timed out after %.0f s (rank %d of %d)", SHM_TIMEOUT_S, rank, nranks);
			}
			usleep(20);
		}
		return SDT_OK;
	}

	// every rank contributes `bytes` (<= SHM_CTRL_BYTES); out receives nranks * bytes, rank order
	int allgather_host(const void *mine, void *out, size_t bytes)
	{
		if (kind == 0 || (kind == 2 && nranks == 1)) {
			memcpy(out, mine, bytes);
			return SDT_OK;
		}
		if (bytes > SHM_CTRL_BYTES)
			return fail(SDT_EINVAL, "control message of %zu bytes", bytes);
		if (kind == 2) {
			memcpy(ctrl(rank), mine, bytes);
			int rc = shm_barrier();
			if (rc != SDT_OK) return rc;
			for (int r = 0; r < nranks; r++)
				memcpy((uint8_t *)out + (size_t)r * bytes, ctrl(r), bytes);
			return shm_barrier();
		}
		uint8_t *d = (uint8_t *)d_ctrl;
		memcpy(h_ctrl, mine, bytes);
		HIPCHK(hipMemcpyAsync(d, h_ctrl, bytes, hipMemcpyHostToDevice, xstream));
		NCCLCHK(g_rccl.AllGather(d, d + SHM_CTRL_BYTES, bytes, NCCL_UINT8, nccl, xstream));
		HIPCHK(hipMemcpyAsync((uint8_t *)h_ctrl + SHM_CTRL_BYTES, d + SHM_CTRL_BYTES, bytes * nranks, hipMemcpyDeviceToHost, xstream));
		{ const int rcw = sync_watched(xstream, "the all-gather of control data"); if (rcw != SDT_OK) return rcw; }
		memcpy(out, (uint8_t *)h_ctrl + SHM_CTRL_BYTES, bytes * nranks);
		return SDT_OK;
	}

	// RCCL collectives have no timeout: a peer that has left keeps this rank's stream waiting for ever.  With a communicator over
	// several ranks every host-side wait therefore polls, and gives up -- with the rank, the place and the exchange round in the
	// message, never by re-executing anything -- when the stream has not completed after a deadline: SDT_COMM_TIMEOUT_S seconds
	// (default 1800) for a wait on the exchange stream -- a collective, which also waits for the SLOWEST peer's work before it (rank 0
	// taking every shard into its table while the others stand in the token all-reduce) --, SDT_DRAIN_TIMEOUT_S (default 7200) for a
	// drain of this rank's own kernels, which needs no peer at all.  The deadline is total time, not time without progress: the
	// stream offers no heartbeat to tell the two apart; a job that legitimately waits longer raises the knob.
	int sync_watched(hipStream_t s, const char *what)
	{
		if (kind != 1 || nranks == 1) {
			HIPCHK(hipStreamSynchronize(s));
			return SDT_OK;
		}
		static const double comm_limit = sdt_env("SDT_COMM_TIMEOUT_S") && atof(sdt_env("SDT_COMM_TIMEOUT_S")) > 0 ? atof(sdt_env("SDT_COMM_TIMEOUT_S")) : 1800.0;
		static const double drain_limit = sdt_env("SDT_DRAIN_TIMEOUT_S") && atof(sdt_env("SDT_DRAIN_TIMEOUT_S")) > 0 ? atof(sdt_env("SDT_DRAIN_TIMEOUT_S")) : 7200.0;
		const double limit = s == xstream ? comm_limit : drain_limit;
		const double t0 = comm_now();
		for (;;) {
			const hipError_t e = hipStreamQuery(s);
			if (e == hipSuccess) return SDT_OK;
			if (e != hipErrorNotReady)
				return fail(SDT_EHIP, "rank %d of %d: %s: %s", rank, nranks, what, hipGetErrorString(e));
			if (comm_now() - t0 > limit)
				return fail(SDT_EHIP, "rank %d of %d: %s not complete after %.0f s (exchange %llu, %llu bytes sent so far): a peer has probably left -- giving up "
				            "(SDT_COMM_TIMEOUT_S / SDT_DRAIN_TIMEOUT_S raise the deadlines)",
				            rank, nranks, what, limit, (unsigned long long)exchanges, (unsigned long long)bytes_sent);
			usleep(50);
		}
	}

	int allreduce_sum_host(int64_t *v, int n)
	{
		if (kind == 0 || (kind == 2 && nranks == 1))
			return SDT_OK;
		std::vector<int64_t> all((size_t)n * nranks);
		int rc = allgather_host(v, all.data(), (size_t)n * sizeof(int64_t));
		if (rc != SDT_OK) return rc;
		for (int i = 0; i < n; i++) {
			int64_t s = 0;
			for (int r = 0; r < nranks; r++) s += all[(size_t)r * n + i];
			v[i] = s;
		}
		return SDT_OK;
	}

	// Move bytes between device buffers: send_ptr[p] / send_bytes[p] go to rank p, recv_ptr[p] / recv_bytes[p] come from
	// rank p (p == rank is skipped: the caller places its own share itself) -- for `nsets` sets of buffers at once (the chunk
	// payloads and their meta words travel in ONE group).  The call returns once the exchange is enqueued on xstream (RCCL) or
	// done (SHM).  peer_outbox_off[set][src * nranks + dst]: SHM layout, computed by the caller from the count matrix.
	int exchange(int nsets, void *const *const *send_ptr, const size_t *const *send_bytes, void *const *const *recv_ptr,
	             const size_t *const *recv_bytes, const size_t *const *peer_outbox_off)
	{
		if (nranks == 1 || kind == 0)
			return SDT_OK;
		size_t sb = 0, rb = 0;
		for (int s = 0; s < nsets; s++)
			for (int p = 0; p < nranks; p++)
				if (p != rank) { sb += send_bytes[s][p]; rb += recv_bytes[s][p]; }
		bytes_sent += sb;
		bytes_recv += rb;
		exchanges++;
		if (kind == 1) {
			if (ev_used == ev.size()) {
				std::pair<hipEvent_t, hipEvent_t> e;
				HIPCHK(hipEventCreate(&e.first));
				HIPCHK(hipEventCreate(&e.second));
				ev.push_back(e);
			}
			HIPCHK(hipEventRecord(ev[ev_used].first, xstream));
			NCCLCHK(g_rccl.GroupStart());
			// pieces of at most 256 MiB, the same on both sides (a rank's send_bytes[p] is p's recv_bytes[rank]): collectives of
			// this RCCL past 1 GiB per call were seen to deliver garbage in round 1, and nothing is lost by staying far below
			const size_t PIECE = (size_t)256 << 20;
			for (int s = 0; s < nsets; s++)
				for (int p = 0; p < nranks; p++) {
					if (p == rank) continue;
					for (size_t o = 0; o < send_bytes[s][p]; o += PIECE)
						NCCLCHK(g_rccl.Send((const char *)send_ptr[s][p] + o, send_bytes[s][p] - o < PIECE ? send_bytes[s][p] - o : PIECE, NCCL_UINT8, p, nccl, xstream));
					for (size_t o = 0; o < recv_bytes[s][p]; o += PIECE)
						NCCLCHK(g_rccl.Recv((char *)recv_ptr[s][p] + o, recv_bytes[s][p] - o < PIECE ? recv_bytes[s][p] - o : PIECE, NCCL_UINT8, p, nccl, xstream));
				}
			NCCLCHK(g_rccl.GroupEnd());
			HIPCHK(hipEventRecord(ev[ev_used].second, xstream));
			ev_used++;
			return SDT_OK;
		}
		// SHM: device -> my outbox, barrier, peers' outboxes -> device, barrier
		if (!xstream)
			return fail(SDT_ESTATE, "shared-memory transport opened without a device");
		HIPCHK(hipStreamSynchronize(xstream));
		const double t0 = comm_now();
		for (int s = 0; s < nsets; s++)
			for (int p = 0; p < nranks; p++) {
				if (p == rank || !send_bytes[s][p]) continue;
				const size_t off = peer_outbox_off[s][(size_t)rank * nranks + p];
				if (off + send_bytes[s][p] > outbox_bytes)
					return fail(SDT_ENOMEM, "shared-memory transport: outbox of %zu MiB too small (set SDT_SHM_OUTBOX_MB)", outbox_bytes >> 20);
				HIPCHK(hipMemcpy(outbox(rank) + off, send_ptr[s][p], send_bytes[s][p], hipMemcpyDeviceToHost));
			}
		int rc = shm_barrier();
		if (rc != SDT_OK) return rc;
		for (int s = 0; s < nsets; s++)
			for (int p = 0; p < nranks; p++) {
				if (p == rank || !recv_bytes[s][p]) continue;
				const size_t off = peer_outbox_off[s][(size_t)p * nranks + rank];
				HIPCHK(hipMemcpy(recv_ptr[s][p], outbox(p) + off, recv_bytes[s][p], hipMemcpyHostToDevice));
			}
		rc = shm_barrier();
		exchange_ms += (comm_now() - t0) * 1e3;
		return rc;
	}

	// RCCL: add the time of the exchanges enqueued since the last call (call after a sync of xstream: all have completed)
	int harvest_time()
	{
		for (; kind == 1 && ev_harvested < ev_used; ev_harvested++) {
			float t = 0;
			if (hipEventElapsedTime(&t, ev[ev_harvested].first, ev[ev_harvested].second) == hipSuccess)
				exchange_ms += t;
		}
		if (ev_harvested == ev_used)
			ev_harvested = ev_used = 0;                  // every pair is free again
		return SDT_OK;
	}

	int barrier()
	{
		int64_t z = 0;
		return allreduce_sum_host(&z, 1);
	}

	int open_rccl(const NcclId *id, int r, int n)
	{
		int rc = rccl_load();
		if (rc != SDT_OK) return rc;
		rank = r; nranks = n;
		HIPCHK(hipStreamCreateWithFlags(&xstream, hipStreamNonBlocking));
		NCCLCHK(g_rccl.CommInitRank(&nccl, n, *id, r));
		HIPCHK(hipMalloc(&d_ctrl, SHM_CTRL_BYTES * (size_t)(n + 1)));
		HIPCHK(hipHostMalloc(&h_ctrl, SHM_CTRL_BYTES * (size_t)(n + 1), hipHostMallocDefault));
		kind = 1;
		return SDT_OK;
	}

	int open_shm(const char *name, int r, int n, bool with_device)
	{
		rank = r; nranks = n;
		if (with_device)
			HIPCHK(hipStreamCreateWithFlags(&xstream, hipStreamNonBlocking));
		const char *mb = sdt_test_env("SDT_SHM_OUTBOX_MB");
		outbox_bytes = (size_t)(mb ? atoll(mb) : 256) << 20;
		shm_bytes = SHM_HEADER_BYTES + (size_t)n * SHM_CTRL_BYTES + (size_t)n * outbox_bytes;
		snprintf(shm_name, sizeof shm_name, "/sdt_%s", name);
		int fd = -1;
		const double t0 = comm_now();
		if (r == 0) {
			shm_unlink(shm_name);
			fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
			if (fd < 0 || ftruncate(fd, (off_t)shm_bytes) != 0)
				return fail(SDT_ENOMEM, "shm_open/ftruncate(%s, %zu MiB) failed", shm_name, shm_bytes >> 20);
		} else {
			for (;;) {
				fd = shm_open(shm_name, O_RDWR, 0600);
				struct stat st;
				if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= shm_bytes)
					break;
				if (fd >= 0) close(fd);
				if (comm_now() - t0 > SHM_TIMEOUT_S)
					return fail(SDT_EHIP, "shared-memory transport: %s did not appear", shm_name);
				usleep(1000);
			}
		}
		shm = (uint8_t *)mmap(nullptr, shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (shm == (uint8_t *)MAP_FAILED) { shm = nullptr; return fail(SDT_ENOMEM, "mmap(%s) failed", shm_name); }
		ShmHeader *h = hdr();
		if (r == 0) {
			h->arrived.store(0); h->generation.store(0); h->failed.store(0);
			h->nranks = (uint32_t)n; h->ctrl_bytes = SHM_CTRL_BYTES; h->outbox_bytes = outbox_bytes;
			h->ready.store(0x5D7C0DE, std::memory_order_release);
		} else {
			while (h->ready.load(std::memory_order_acquire) != 0x5D7C0DE) {
				if (comm_now() - t0 > SHM_TIMEOUT_S)
					return fail(SDT_EHIP, "shared-memory transport: rank 0 never initialised %s", shm_name);
				usleep(1000);
			}
			if (h->nranks != (uint32_t)n || h->outbox_bytes != outbox_bytes)
				return fail(SDT_EINVAL, "shared-memory transport: ranks disagree about the geometry of %s", shm_name);
		}
		kind = 2;
		return shm_barrier();
	}

	void close_all()
	{
		if (kind == 1 && nccl) (void)g_rccl.CommDestroy(nccl);
		if (kind == 2 && shm) {
			munmap(shm, shm_bytes);                  // (no barrier: a rank that failed must not hold the others; the name goes with rank 0)
			if (rank == 0) shm_unlink(shm_name);
		}
		if (d_ctrl) (void)hipFree(d_ctrl);
		if (h_ctrl) (void)hipHostFree(h_ctrl);
		for (auto &e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
		if (xstream) (void)hipStreamDestroy(xstream);
		*this = Comm();
	}
};

// rank that owns level-1 bucket b: contiguous ranges, sizes differ by at most one
__host__ __device__ inline int sk_owner_of_bucket(uint32_t b1, int nranks) { return (int)((b1 * (uint32_t)nranks) >> SK_L1BITS); }
// first bucket of rank r
inline uint32_t sk_first_bucket(int r, int nranks)
{
	uint32_t b = (uint32_t)(((uint64_t)r << SK_L1BITS) / (uint32_t)nranks);
	while (b < (uint32_t)SK_NB1 && sk_owner_of_bucket(b, nranks) < r) b++;
	while (b > 0 && sk_owner_of_bucket(b - 1, nranks) >= r) b--;
	return b;
}

} // namespace sdt
