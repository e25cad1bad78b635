"""Owner-computes sharding across GPUs: the exchange step of the path.

The reference routes every record to thread `hash_kmer % thrd_num` by letting each thread scan the whole
batch (prlHashReads.c:77-90).  Across GPUs the owner of a canonical k-mer is
    owner = ((sdt_owner_hash(key) >> 32) * nranks) >> 32
and records travel once, in an all-to-all(v) over xGMI (RCCL; `gloo` in the CPU tests).  ROUND-1 PATH, kept for A/B
(`bench.py --shard-mode route`): the product path is the C-level bucket sharding of include/sdt_gpu.h
(sdt_gpu_count_reads_sharded), which moves ~3 B per k-mer instead of 16.  This module is the
device-agnostic plumbing around the two kernels (`sdt_gpu_extract_route`, `sdt_gpu_insert_records`): it only
moves opaque 8-byte words with torch.distributed.
"""
from __future__ import annotations

import numpy as np

# Measured on MI355X / RCCL 2.26 / torch 2.10 (one rank): all_to_all_single silently delivers garbage once the
# message exceeds 2^27 int64 elements (1 GiB) -- tools/dbg_exchange.py.  Whether the limit is per split or per call
# could not be told apart on one GPU, so the WHOLE call is kept below half of it, and anything larger fails loudly.
MAX_CALL_WORDS = 1 << 26


def owner_of(lib, key_words_msw_first: np.ndarray, nranks: int) -> int:
    """host copy of the device owner function (one key: uint64[nw], most significant word first)"""
    a = np.ascontiguousarray(key_words_msw_first, dtype=np.uint64)
    h = lib.sdt_owner_hash(a.ctypes.data, a.size)
    return ((h >> 32) * nranks) >> 32


def exchange_records(send, counts, cap_per_rank: int, rec_words: int, recv, group=None, verify: bool = False):
    """all-to-all(v) of routed records.

    send   : int64 tensor, nranks slices of cap_per_rank records (rec_words int64 each); slice r holds
             counts[r] valid records destined to rank r (layout written by sdt_gpu_extract_route)
    counts : int64 tensor [nranks] (same device as send)
    recv   : int64 tensor with room for everything this rank receives
    Returns (number of records received, per-source counts list)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rcounts = torch.empty_like(counts)
    dist.all_to_all_single(rcounts, counts, group=group)
    c = counts.cpu().tolist()
    rc = rcounts.cpu().tolist()
    # k_extract_route keeps counting past a full slice (the records themselves are dropped and flagged): a slice that
    # overflowed would make `send[i*cap : i*cap + c[i]]` reach into the next owner's slice.  Every rank must learn of it
    # in the same place, or the others wait in the next collective until it times out.
    over = torch.tensor([1 if max(c) > cap_per_rank else 0], dtype=torch.int64, device=counts.device)
    dist.all_reduce(over, op=dist.ReduceOp.MAX, group=group)
    if int(over.item()):
        raise RuntimeError(f"an owner's slice overflowed (most records for one owner on this rank: {max(c)}, capacity {cap_per_rank}): "
                           "use fewer reads per round (ShardedCounter sizes the slices at 1.25 x the mean)")
    total = int(sum(rc))
    if total * rec_words > recv.numel():
        raise RuntimeError(f"receive buffer too small: need {total} records, have {recv.numel() // rec_words}")
    if max(sum(c), sum(rc)) * rec_words > MAX_CALL_WORDS:
        raise RuntimeError(f"all-to-all of {max(sum(c), sum(rc)) * rec_words} words exceeds {MAX_CALL_WORDS}: "
                           "use fewer reads per round")
    # compact the fixed-capacity slices (cheap device copy), then ONE all-to-all(v) of 8-byte words
    ins = [send[(i * cap_per_rank) * rec_words:(i * cap_per_rank + c[i]) * rec_words] for i in range(world)]
    packed = torch.cat(ins) if world > 1 else ins[0]
    dist.all_to_all_single(recv[: total * rec_words], packed,
                           output_split_sizes=[x * rec_words for x in rc],
                           input_split_sizes=[x * rec_words for x in c], group=group)
    if verify:
        # every word sent somewhere must arrive somewhere: compare the global wrap-around sums
        chk = torch.stack([packed.sum(), recv[: total * rec_words].sum()])
        dist.all_reduce(chk, group=group)
        if int(chk[0].item()) != int(chk[1].item()):
            raise RuntimeError("all-to-all checksum mismatch: records were lost or corrupted in the exchange")
    return total, rc


def allreduce_stats(hist: np.ndarray, kmers: int, nodes: int, linear: int, device, group=None):
    """sum the per-rank 257-bin kmerFreq histogram and counters (the reference sums per-thread bins in
    freqStat, prlHashReads.c:1004-1014)"""
    import torch
    import torch.distributed as dist

    h = torch.from_numpy(np.ascontiguousarray(hist, dtype=np.int64)).to(device)
    agg = torch.tensor([kmers, nodes, linear], dtype=torch.int64, device=device)
    dist.all_reduce(h, group=group)
    dist.all_reduce(agg, group=group)
    k, n, l = (int(x) for x in agg.cpu().tolist())
    return h.cpu().numpy(), k, n, l


class ShardedCounter:
    """pass 1 on `world` ranks: each rank owns one PregraphGPU shard and a slice of the reads"""

    def __init__(self, g, world: int, max_read_len: int, reads_per_round: int, device):
        import torch

        self.g, self.world, self.L, self.device = g, world, max_read_len, device
        self.rec_words = g.record_bytes() // 8
        per_read = max(max_read_len - g.K + 1, 1)
        # keep every exchange below the collective's safe size (with head room: what a rank RECEIVES can exceed
        # what it sends when owners are unevenly loaded)
        limit = int(MAX_CALL_WORDS / self.rec_words / 1.3 / per_read)
        self.per_round = max(64, min(reads_per_round, limit))
        reads_per_round = self.per_round
        kmers_round = reads_per_round * per_read
        self.cap = int(kmers_round / world * 1.25) + 4096
        n = self.cap * world * self.rec_words
        self.send = torch.empty(n, dtype=torch.int64, device=device)
        self.recv = torch.empty(n, dtype=torch.int64, device=device)
        self.counts = torch.zeros(world, dtype=torch.int64, device=device)
        self.displs = torch.zeros(world, dtype=torch.int64, device=device)

    def count_reads(self, words, nwords: int, offsets, nreads: int, group=None, verify: bool = False):
        g = self.g
        for r0 in range(0, nreads, self.per_round):
            nr = min(self.per_round, nreads - r0)
            g.extract_route(words, nwords, offsets[r0:], nr, self.L, self.world, self.send, self.cap * self.world,
                            self.counts, self.displs)
            total, _ = exchange_records(self.send, self.counts, self.cap, self.rec_words, self.recv, group, verify)
            g.insert_records(self.recv, total)
