// sdt_ctg_kernels.cuh -- the `map` stage on the device:
//   k_index_contigs : prlContig2nodes (prlHashCtg.c:287-425) -- every k-mer of every contig into the node table;
//                     the FIRST occurrence (contig order, then position) owns the payload, any further
//                     occurrence only marks the node deleted (singleKmer :110-139)
//   k_align_reads   : prlRead2Ctg's chop + searchKmer + parse1read (prlRead2Ctg.c:129-353) -- one wavefront per read
//
// The index reuses the pass-1 table unchanged: `count` says whether a k-mer is unique (deleted <=> count > 1) and
// the first-occurrence slot (Table::first, atomicMin) carries  contig ordinal << 25 | position << 1 | smaller,
// so the result does not depend on the order in which lanes get to a slot.
#pragma once
#include "sdt_table.cuh"

namespace sdt {

constexpr int CTG_POS_BITS = 24;                         // kmer_t.r_links is a 24-bit field (newhash.h:65-77)
constexpr int MAX_HITS = 20;                             // pos_temp[20], prlRead2Ctg.c:240

struct Hit {                                             // == sdt_hit of include/sdt_gpu.h
	uint32_t contig;
	int32_t contig_offset;
	uint32_t read_offset;
	uint32_t align_len_orien;                            // k-mers on that contig | ('-' ? 1u << 31 : 0)
};

// 64 bits of the packed stream starting at stream bit `bit` (bit 0 = top bit of word 0); bits before the stream
// read as zero.  The stream is followed by >= 4 pad words.
__device__ inline uint64_t stream_bits64(const uint32_t *__restrict__ w, int64_t bit)
{
	const int64_t wi = bit >> 5;                         // floor
	const int sh = (int)(bit & 31);
	const uint64_t a = wi >= 0 ? w[wi] : 0u, b = wi + 1 >= 0 ? w[wi + 1] : 0u, c = wi + 2 >= 0 ? w[wi + 2] : 0u;
	const uint64_t hi = (a << 32) | b;
	return sh ? ((hi << sh) | (c >> (32 - sh))) : hi;
}

// the K bases starting at base index p of a packed stream in GLOBAL memory, right-aligned big-endian
template <int NW> __device__ inline Key<NW> global_kmer(const uint32_t *__restrict__ w, uint64_t p, int K)
{
	Key<NW> k;
	const int64_t e = 2 * (int64_t)(p + (uint64_t)K);    // one past the last bit of the k-mer
#pragma unroll
	for (int i = 0; i < NW; i++) {
		k.w[i] = stream_bits64(w, e - 64 * (int64_t)(NW - i));
		const int bits = 2 * K - 64 * (NW - 1 - i);
		if (bits <= 0) k.w[i] = 0;
		else if (bits < 64) k.w[i] &= (1ULL << bits) - 1ULL;
	}
	return k;
}

// one lane per base position of the batch; contig c = the one whose [offsets[c], offsets[c+1]) holds it
template <int NW>
__global__ __launch_bounds__(TPB) void k_index_contigs(const uint32_t *__restrict__ words, const uint64_t *__restrict__ offs,
                                                       uint64_t nctg, uint64_t ord_base, int K, Table<NW> tbl, Stats *stats)
{
	const uint64_t total = offs[nctg];
	uint32_t claimed = 0, failed = 0, done = 0;
	for (uint64_t p = blockIdx.x * (uint64_t)TPB + threadIdx.x; p < total; p += (uint64_t)gridDim.x * TPB) {
		uint64_t lo = 0, hi = nctg;                      // largest c with offs[c] <= p
		while (hi - lo > 1) {
			const uint64_t mid = (lo + hi) >> 1;
			if (offs[mid] <= p) lo = mid; else hi = mid;
		}
		const uint64_t start = offs[lo], len = offs[lo + 1] - start, j = p - start;
		if (len < (uint64_t)K || j > len - (uint64_t)K) continue;
		const Key<NW> fw = global_kmer<NW>(words, p, K);
		const Key<NW> rc = key_revcomp<NW>(fw, K);
		const bool smaller = key_less<NW>(fw, rc);       // KmerSmaller(word, bal_word), prlHashCtg.c:205
		const uint64_t ord = ((ord_base + lo) << (CTG_POS_BITS + 1)) | ((j & ((1ULL << CTG_POS_BITS) - 1)) << 1) | (uint64_t)smaller;
		if (table_put<NW>(tbl, smaller ? fw : rc, 4u, 4u, claimed, ord)) done++;
		else failed++;
	}
	if (done) atomicAdd(&stats->kmers, (unsigned long long)done);
	if (claimed) atomicAdd(&stats->distinct, (unsigned long long)claimed);
	if (failed) atomicAdd(&stats->probe_fail, (unsigned long long)failed);
}

// After the last contig: turn every node into what a look-up needs, in place -- val = PL_VALID | contig id |
// position << 32 | smaller_in_contig << 56 for a k-mer that occurs once, 0 for a repeated ("deleted") one -- so that
// searchKmer is ONE 16..48-byte entry read instead of entry + count high word + first-occurrence word.
constexpr uint64_t PL_VALID = 1ULL << 63;

template <int NW>
__global__ __launch_bounds__(TPB) void k_finalize_contig_index(Table<NW> tbl, const uint32_t *__restrict__ ctg_ids, uint64_t n_ord, Stats *stats)
{
	const uint64_t slots = tbl.slots();
	uint32_t bad = 0;
	for (uint64_t s = blockIdx.x * (uint64_t)TPB + threadIdx.x; s < slots; s += (uint64_t)gridDim.x * TPB) {
		if (tbl.ent[s].key[0] == KEY_EMPTY) continue;
		const uint64_t cnt = (tbl.ent[s].val >> 48) | ((uint64_t)(tbl.aux[s] & 0xFFFFu) << 16);
		uint64_t v = 0;
		if (cnt == 1) {                                  /* found && !node->deleted */
			const uint64_t f = tbl.first[s];
			const uint64_t ord = f >> (CTG_POS_BITS + 1);
			if (ord < n_ord) v = PL_VALID | (uint64_t)ctg_ids[ord] | (((f >> 1) & ((1ULL << CTG_POS_BITS) - 1)) << 32) | ((f & 1ULL) << 56);
			else bad++;
		}
		tbl.ent[s].val = v;
	}
	if (bad) atomicAdd(&stats->probe_fail, (unsigned long long)bad);
}

template <int NW> __device__ inline bool lookup_slot(const Table<NW> &tbl, const Key<NW> &k, uint64_t &slot_out)
{
	return table_find<NW>(tbl, k, slot_out);
}

// searchKmer on the finalized index: the node's val (0 = absent or deleted)
template <int NW> __device__ inline uint64_t lookup_val(const Table<NW> &tbl, const Key<NW> &k)
{
	uint64_t slot, lo, n;
	probe_begin<NW>(tbl, k, slot, lo, n);
	for (uint64_t probe = 0; probe < n; probe++, slot = probe_next(slot, lo, n)) {
		if constexpr (NW == 1) {
			const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(tbl.ent + slot);     // key + val in one 16-byte load
			if (kv.x == KEY_EMPTY) return 0;
			if (kv.x == k.w[0]) return kv.y;
		} else {
			const Entry<NW> *e = tbl.ent + slot;
			if (e->key[0] == KEY_EMPTY) return 0;
			bool same = true;
#pragma unroll
			for (int w = 0; w < NW; w++) same = same && e->key[w] == k.w[w];
			if (same) return e->val;
		}
	}
	return 0;
}

// LDS word per k-mer of the read: the node's finalized val | smaller_in_read << 57

// read_info[r] = hit_start (40 bits) | nhits << 40 | best << 48 | footprint << 56 | overflow << 57
template <int NW>
__global__ __launch_bounds__(TPB) void k_align_reads(const uint32_t *__restrict__ words, const uint64_t *__restrict__ offs, uint64_t nreads,
                                                     const int32_t *__restrict__ align_len, int align_len_all, int K, Table<NW> tbl,
                                                     const uint32_t *__restrict__ ctg_len, const uint32_t *__restrict__ ctg_twin, uint64_t num_ctg,
                                                     int max_kmers, int waves_per_block, uint64_t *__restrict__ read_info, Hit *__restrict__ hits,
                                                     unsigned long long max_hits, unsigned long long *hit_cursor, Stats *stats)
{
	extern __shared__ uint64_t smem[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (wave >= waves_per_block) return;
	uint64_t *pl = smem + (size_t)wave * (size_t)(max_kmers + 2 * MAX_HITS);
	Hit *my_hits = reinterpret_cast<Hit *>(pl + max_kmers);          // 20 x 16 B
	uint32_t bad = 0;
	for (uint64_t r = blockIdx.x * (uint64_t)waves_per_block + wave; r < nreads; r += (uint64_t)gridDim.x * waves_per_block) {
		const uint64_t start = offs[r];
		const int len = (int)(offs[r + 1] - start);
		if (len < K + 1 || len - K + 1 > max_kmers) {                  // too short (prlRead2Ctg.c:133-136) -- or longer than promised
			if (len >= K + 1) bad++;
			if (lane == 0) read_info[r] = 0;
			continue;
		}
		const int n = len - K + 1;
		// chopKmer4read + searchKmer
		for (int j = lane; j < n; j += 64) {
			const Key<NW> fw = global_kmer<NW>(words, start + (uint64_t)j, K);
			const Key<NW> rc = key_revcomp<NW>(fw, K);
			const bool smaller = key_less<NW>(fw, rc);
			uint64_t v = lookup_val<NW>(tbl, smaller ? fw : rc);
			if (v) v |= (uint64_t)smaller << 57;
			pl[j] = v;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// parse1read: group the found k-mers by contig in order of first appearance
		int al = align_len ? align_len[r] : align_len_all;
		const int alldgn = len > al ? al : len;
		const int multi = alldgn - K + 1 < 5 ? 5 : alldgn - K + 1;
		int nh = 0, counter2 = 0, overflow = 0;
		uint32_t best_key = 0;                                           // count << 8 | (255 - hit index): max = most k-mers, first such
		// Groups in order of first appearance (the reference's outer loop over j with its NULL-ing of later k-mers of
		// the same contig, :258-291): the leader is the first k-mer still standing; one ballot per 64 positions counts
		// its contig's k-mers and takes them out.  Everything below is uniform across the wavefront.
		for (int lead_base = 0; lead_base < n;) {
			const int lj = lead_base + lane;
			const uint64_t lv = lj < n ? pl[lj] : 0;
			const unsigned long long lm = __ballot((lv & PL_VALID) != 0);
			if (!lm) { lead_base += 64; continue; }
			const int ll = __ffsll((long long)lm) - 1;
			const int j = lead_base + ll;                                   // first k-mer of the group
			const uint64_t me = __shfl(lv, ll);
			const uint32_t ctg = (uint32_t)me;
			int cnt = 0;
			for (int base = lead_base; base < n; base += 64) {
				const int s = base + lane;
				const uint64_t o = s < n ? pl[s] : 0;
				const bool same = (o & PL_VALID) && (uint32_t)o == ctg;
				cnt += __popcll(__ballot(same));
				if (same) pl[s] = 0;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();
			counter2 += cnt >= 2;
			if (cnt < multi) continue;
			if (nh < MAX_HITS && lane == 0) {
				const uint32_t pos = (uint32_t)(me >> 32) & 0xFFFFFFu;
				const bool sm_ctg = (me >> 56) & 1ULL, sm_read = (me >> 57) & 1ULL;
				Hit h;
				h.read_offset = (uint32_t)j + 1u;
				if (ctg > num_ctg) {
					bad++;
					h.contig = 0; h.contig_offset = 0; h.align_len_orien = (uint32_t)cnt;
				} else if (sm_ctg != sm_read) {                               // node->twin == isSmaller  (twin = !smaller_in_contig)
					h.contig = ctg_twin[ctg];
					h.contig_offset = (int32_t)(ctg_len[ctg] - pos - (uint32_t)K);
					h.align_len_orien = (uint32_t)cnt | (1u << 31);
				} else {
					h.contig = ctg;
					h.contig_offset = (int32_t)pos;
					h.align_len_orien = (uint32_t)cnt;
				}
				my_hits[nh] = h;
			}
			if (nh < MAX_HITS) {
				const uint32_t key = ((uint32_t)cnt << 8) | (uint32_t)(255 - nh);
				if (key > best_key) best_key = key;
			}
			nh++;
		}
		if (nh > MAX_HITS) { overflow = 1; nh = 0; }                     // the reference writes past pos_temp[20] here: undefined
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// hits[r] = the read's first hit; further hits (rare) go to the tail of hits[], handed out by one shared cursor
		// that starts at nreads -- a cursor bump per READ would serialise at the ~80 M/s of same-address atomics
		unsigned long long hstart = 0;
		if (nh) {
			if (lane == 0 && r < max_hits) hits[r] = my_hits[0];
			if (nh > 1) {
				if (lane == 0) hstart = atomicAdd(hit_cursor, (unsigned long long)(nh - 1));
				hstart = __shfl(hstart, 0);
				if (lane >= 1 && lane < nh && hstart + (unsigned long long)(lane - 1) < max_hits) hits[hstart + lane - 1] = my_hits[lane];
			}
		}
		if (lane == 0) {
			const uint64_t best = nh ? (uint64_t)(255 - (best_key & 255u)) : 0;
			read_info[r] = nh ? ((hstart & ((1ULL << 40) - 1)) | ((uint64_t)nh << 40) | (best << 48) | ((uint64_t)(counter2 > 1) << 56))
			                  : ((uint64_t)overflow << 57);
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	if (bad) atomicAdd(&stats->probe_fail, (unsigned long long)bad);
}

} // namespace sdt
