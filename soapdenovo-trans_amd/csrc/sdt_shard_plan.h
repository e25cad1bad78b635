// sdt_shard_plan.h -- the exchange plan of the multi-GPU path as PURE host functions of the count matrix.
//
// After a round's level-1 scatter every rank all-gathers its chunk list offsets per bucket: mat[r * 257 + b] = first position of
// bucket b in rank r's chunk list (mat[r * 257 + 256] = its length).  Everything a rank needs to lay out its own and every
// peer's buffers follows from that matrix and the bucket ranges (rank d owns buckets [ranges[d], ranges[d + 1])) -- no second
// message.  The functions below ARE that computation: sk_flush_sharded (sdt_gpu.hip) calls them, and so does the CPU test
// that drives the protocol with two gloo ranks (tests/test_multirank_gloo.py), through sdt_shard_plan / sdt_shard_cut_ranges.
// Reference: prlHashReads.c:77-90 routes a record to thread `hash_kmer % thrd_num`; here the unit is a chunk of super-k-mer
// records and the owner is the rank that owns the record's minimizer bucket.
#pragma once
#include <stdint.h>

namespace sdt {

#ifndef SDT_SK_L1BITS
#define SDT_SK_L1BITS 8
#endif
constexpr int SHARD_NB1 = 1 << SDT_SK_L1BITS;                   // level-1 buckets (SK_NB1; static_assert in sdt_gpu.hip)
constexpr int SHARD_MAX_RANKS = 64;

inline uint32_t shard_mat(const uint32_t *mat, int r, uint32_t b) { return mat[(size_t)r * (SHARD_NB1 + 1) + b]; }

// Ownership: contiguous ranges of the level-1 buckets, weight = chunks of the bucket summed over the ranks (+1: an empty sample still
// gives every bucket a weight); every rank owns at least one bucket.  ranges[0..n].
// The cut MINIMISES THE HEAVIEST RANGE (round 6): the job ends when the slowest rank does, and with N = 8 a rank owns ~32 buckets of
// very unequal weight (expression skew: one minimizer of a highly expressed transcript can be several per cent of all records).
// Rounds 2-5 cut greedily at the first bucket past r / n of the total, which lets one range overshoot by a whole bucket.  Here: the
// smallest capacity C for which a greedy fill needs at most n ranges (bisection over C -- the classic linear partition), then that
// fill.  Pure function of the matrix: every rank computes the same ranges.
inline void shard_cut_ranges(const uint32_t *mat, int n, uint32_t *ranges)
{
	uint64_t w[SHARD_NB1], total = 0, wmax = 0;
	for (uint32_t b = 0; b < (uint32_t)SHARD_NB1; b++) {
		uint64_t wgt = 1;
		for (int r = 0; r < n; r++)
			wgt += shard_mat(mat, r, b + 1) - shard_mat(mat, r, b);
		w[b] = wgt;
		total += wgt;
		if (wgt > wmax) wmax = wgt;
	}
	auto parts = [&](uint64_t cap) {
		int k = 1;
		uint64_t sum = 0;
		for (uint32_t b = 0; b < (uint32_t)SHARD_NB1; b++) {
			if (sum + w[b] > cap) { k++; sum = 0; }
			sum += w[b];
		}
		return k;
	};
	uint64_t lo = wmax, hi = total;
	while (lo < hi) {
		const uint64_t mid = lo + (hi - lo) / 2;
		if (parts(mid) <= n) hi = mid; else lo = mid + 1;
	}
	ranges[0] = 0;
	uint32_t b = 0;
	for (int r = 0; r < n; r++) {
		const uint32_t start = b, limit = (uint32_t)SHARD_NB1 - (uint32_t)(n - r - 1);      // (the ranks behind this one get a bucket each at least)
		uint64_t sum = 0;
		while (b < limit && (b == start || sum + w[b] <= lo)) sum += w[b++];
		ranges[r + 1] = b;
	}
	ranges[n] = SHARD_NB1;
}

// Sub-rounds of one exchange: a rank that would receive more than its buffer holds is visible to EVERY rank in the matrix, so
// all split the exchange the same way and nobody hangs.  (+1 when split: pieces are cut by source, not by size.)
inline uint32_t shard_subrounds(const uint32_t *mat, int n, const uint32_t *ranges, uint32_t recv_chunks)
{
	uint32_t S = 1;
	for (int d = 0; d < n; d++) {
		uint64_t in = 0;
		for (int s = 0; s < n; s++)
			in += shard_mat(mat, s, ranges[d + 1]) - shard_mat(mat, s, ranges[d]);
		const uint32_t need = (uint32_t)((in + recv_chunks - 1) / recv_chunks);
		if (need > S) S = need;
	}
	return S > 1 ? S + 1 : S;
}

// the piece (source s -> destination d) of sub-round t: positions [lo, hi) of s's chunk list
inline void shard_piece(const uint32_t *mat, const uint32_t *ranges, int s, int d, uint32_t t, uint32_t S, uint32_t &lo, uint32_t &hi)
{
	const uint64_t a = shard_mat(mat, s, ranges[d]), b = shard_mat(mat, s, ranges[d + 1]);
	lo = (uint32_t)(a + (b - a) * t / S);
	hi = (uint32_t)(a + (b - a) * (t + 1) / S);
}

// what rank `me` does in sub-round t
struct ShardRound {
	uint32_t send_begin[SHARD_MAX_RANKS];        // piece for rank p: starts here in my chunk list ...
	uint32_t send_count[SHARD_MAX_RANKS];        // ... and is this long
	uint32_t send_at[SHARD_MAX_RANKS];           // p != me: its first chunk in my send buffer; p == me: in my RECEIVE buffer
	uint32_t recv_count[SHARD_MAX_RANKS];        // run that arrives from rank s ...
	uint32_t recv_at[SHARD_MAX_RANKS];           // ... and where it lies in my receive buffer (runs in rank order, mine included)
	uint32_t send_total, recv_total;             // chunks in the send buffer / in the receive buffer
};

inline void shard_round(const uint32_t *mat, int n, int me, const uint32_t *ranges, uint32_t t, uint32_t S, ShardRound &r)
{
	r.send_total = r.recv_total = 0;
	for (int s = 0; s < n; s++) {
		uint32_t lo, hi;
		shard_piece(mat, ranges, s, me, t, S, lo, hi);
		r.recv_count[s] = hi - lo;
		r.recv_at[s] = r.recv_total;
		r.recv_total += hi - lo;
	}
	for (int p = 0; p < n; p++) {
		uint32_t lo, hi;
		shard_piece(mat, ranges, me, p, t, S, lo, hi);
		r.send_begin[p] = lo;
		r.send_count[p] = hi - lo;
		if (p == me) {
			r.send_at[p] = r.recv_at[me];
		} else {
			r.send_at[p] = r.send_total;
			r.send_total += hi - lo;
		}
	}
}

} // namespace sdt
