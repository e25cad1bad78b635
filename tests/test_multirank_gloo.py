"""CPU, world_size 2 and 3, gloo: the PRODUCT protocol of the multi-GPU path (sdt_gpu_count_reads_sharded), driven through the
very functions the library plans its exchange with -- sdt_shard_cut_ranges / sdt_shard_plan (csrc/sdt_shard_plan.h: pure
host functions of the all-gathered count matrix, no device) -- with torch.distributed as the transport:

  every rank holds a chunk list sorted by level-1 bucket (here: chunks are tagged integers instead of 32 super-k-mer records),
  all-gathers its 257 list offsets, cuts the bucket ranges, and for every sub-round gathers its pieces into a send buffer,
  exchanges them (isend / irecv over gloo = the grouped ncclSend / ncclRecv of the library) and lays the arrivals out in its
  receive buffer exactly where the plan says.

Checked: all ranks agree on the ranges and on the number of sub-rounds without another message; what rank s plans to send to
d is what d plans to receive from s; no sub-round overfills a receive buffer; every chunk of every bucket arrives exactly
once, on the rank that owns its bucket, and the runs of a receive buffer are in bucket order (the level-2 work items of
sk_flush_sharded rely on that).  The kernels either side of the exchange need a GPU: tests/test_sharded.py runs the same
protocol end to end with 2-4 ranks sharing the box's GPU.  Reference: prlHashReads.c:77-90 (records routed by hash_kmer %
thrd_num)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, recv_chunks, seed, out_q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import __graft_entry__ as ge
    pkg = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # this rank's level-1 chunk list: skewed bucket sizes (a few giant minimizers), some buckets empty
        rng = np.random.default_rng(seed + 17 * rank)
        sizes = (rng.pareto(1.2, size=256) * 20).astype(np.int64)
        sizes[rng.integers(0, 256, size=30)] = 0
        off = np.zeros(257, dtype=np.uint32)
        off[1:] = np.cumsum(sizes)
        bucket_of = np.repeat(np.arange(256), sizes)
        # a chunk = one tagged integer: source rank << 40 | bucket << 28 | serial within the bucket
        serial = np.concatenate([np.arange(n) for n in sizes]) if off[256] else np.zeros(0, dtype=np.int64)
        chunks = (rank << 40) | (bucket_of.astype(np.int64) << 28) | serial.astype(np.int64)
        # 1. all-gather of the offsets: the count matrix
        mat_t = [torch.zeros(257, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(mat_t, torch.from_numpy(off.astype(np.int64)))
        mat = np.stack([t.numpy() for t in mat_t]).astype(np.uint32)
        # 2. ranges and sub-rounds: the same on every rank, from the matrix alone
        ranges = pkg.shard_cut_ranges(mat, world)
        assert ranges[0] == 0 and ranges[world] == 256 and (np.diff(ranges.astype(np.int64)) >= 1).all()
        S = pkg.shard_plan(mat, world, rank, ranges, recv_chunks, 0)[0]
        agree = [None] * world
        dist.all_gather_object(agree, (ranges.tolist(), int(S)))
        assert all(a == agree[0] for a in agree), "ranks disagree about ranges / sub-rounds"
        got = []
        for t in range(S):
            S2, sbeg, scnt, sat, rcnt, rat = pkg.shard_plan(mat, world, rank, ranges, recv_chunks, t)
            assert S2 == S
            # what I plan to send to p is what p plans to receive from me
            plans = [None] * world
            dist.all_gather_object(plans, (scnt.tolist(), rcnt.tolist()))
            for p in range(world):
                assert plans[p][1][rank] == scnt[p], (t, rank, p)
                assert plans[rank][1][p] == plans[p][0][rank]
            assert int(rcnt.sum()) <= recv_chunks, "a sub-round overfills the receive buffer"
            # gather: per destination one contiguous run of my chunk list (k_sk_gather); my own share goes straight into
            # my receive buffer at send_at[me]
            send_total = int(sum(int(scnt[p]) for p in range(world) if p != rank))
            sendbuf = np.full(send_total, -1, dtype=np.int64)
            recvbuf = np.full(int(rcnt.sum()), -1, dtype=np.int64)
            for p in range(world):
                piece = chunks[int(sbeg[p]): int(sbeg[p]) + int(scnt[p])]
                if p == rank:
                    recvbuf[int(sat[p]): int(sat[p]) + len(piece)] = piece
                else:
                    sendbuf[int(sat[p]): int(sat[p]) + len(piece)] = piece
            assert (sendbuf >= 0).all()                  # the pieces tile the send buffer
            # exchange (the library: one group of ncclSend / ncclRecv per peer)
            ins = [torch.from_numpy(sendbuf[int(sat[p]): int(sat[p]) + int(scnt[p])].copy()) if p != rank else torch.zeros(0, dtype=torch.int64)
                   for p in range(world)]
            outs = [torch.zeros(int(rcnt[s]) if s != rank else 0, dtype=torch.int64) for s in range(world)]
            reqs = []                                    # point to point, as the library does it (gloo has no all-to-all)
            for p in range(world):
                if p == rank:
                    continue
                if ins[p].numel():
                    reqs.append(dist.isend(ins[p], p))
                if outs[p].numel():
                    reqs.append(dist.irecv(outs[p], p))
            for r_ in reqs:
                r_.wait()
            for s in range(world):
                if s != rank:
                    recvbuf[int(rat[s]): int(rat[s]) + int(rcnt[s])] = outs[s].numpy()
            assert (recvbuf >= 0).all()                  # the runs tile the receive buffer
            for s in range(world):                       # every run comes from ONE source, in bucket order, buckets I own
                run = recvbuf[int(rat[s]): int(rat[s]) + int(rcnt[s])]
                assert ((run >> 40) == s).all()
                b = (run >> 28) & 0xFFF
                assert (np.diff(b) >= 0).all() and ((b >= ranges[rank]) & (b < ranges[rank + 1])).all()
            got.append(recvbuf)
        mine = np.sort(np.concatenate(got)) if got else np.zeros(0, dtype=np.int64)
        # every chunk of my buckets, from every rank, exactly once
        everything = [None] * world
        dist.all_gather_object(everything, chunks.tolist())
        want = np.sort(np.array([c for lst in everything for c in lst
                                 if ranges[rank] <= ((c >> 28) & 0xFFF) < ranges[rank + 1]], dtype=np.int64))
        assert len(mine) == len(want) and (mine == want).all()
        out_q.put((rank, "ok", int(S), len(mine)))
    except Exception as e:      # noqa: BLE001 -- report to the parent instead of hanging the peers
        import traceback
        out_q.put((rank, "fail: " + repr(e) + "\n" + traceback.format_exc(), 0, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,recv_chunks", [(2, 1 << 30), (2, 2500), (3, 1500)])
def test_product_exchange_protocol_over_gloo(world, recv_chunks):
    """one exchange (unsplit, and split into sub-rounds by a small receive buffer) between `world` CPU processes"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, recv_chunks, 1234, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    S = {r[2] for r in res}
    assert len(S) == 1
    if recv_chunks < (1 << 30):
        assert S.pop() > 1, "the small receive buffer must force sub-rounds"
    assert sum(r[3] for r in res) > 0


def test_plan_rejects_bad_arguments():
    import __graft_entry__ as ge
    pkg = ge.load_package()
    mat = np.zeros((2, 257), dtype=np.uint32)
    with pytest.raises(pkg.SdtError):
        pkg.shard_plan(mat, 2, 2, np.array([0, 128, 256], dtype=np.uint32), 10, 0)        # rank out of range
    with pytest.raises(pkg.SdtError):
        pkg.shard_plan(mat, 2, 0, np.array([0, 128, 256], dtype=np.uint32), 10, 5)        # sub-round past the last
    r = pkg.shard_cut_ranges(mat, 2)
    assert r.tolist() == [0, 128, 256]                   # empty sample: equal ranges


@pytest.mark.parametrize("nranks", [2, 3, 4, 8, 16])
def test_cut_ranges_minimise_the_heaviest_range(nranks):
    """sdt_shard_cut_ranges against a dynamic programme over all contiguous partitions of the 256 buckets: the heaviest range is the
    smallest any partition allows (the job ends with its slowest rank), every rank owns a bucket, the ranges tile 0..256 -- for flat,
    log-normal (expression skew) and one-giant-bucket weights.  Round 5's greedy cut (first bucket past r / n of the total) is also
    restated here: it must never beat the library's cut.  Reference: prlHashReads.c:79-88 (`hash_kmer % thrd_num`: the reference
    balances by hashing single k-mers; here the unit of ownership is a minimizer bucket)."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    rng = np.random.default_rng(nranks)
    for trial in range(12):
        kind = trial % 4
        if kind == 0:
            w = np.ones(256, dtype=np.int64) * 7
        elif kind == 1:
            w = np.maximum(1, rng.lognormal(3.0, 2.0, 256)).astype(np.int64)
        elif kind == 2:
            w = rng.integers(1, 50, 256).astype(np.int64)
            w[rng.integers(0, 256)] = int(w.sum() // 3)           # one giant minimizer: a third of everything
        else:
            w = rng.integers(0, 3, 256).astype(np.int64)         # many empty buckets
        # the matrix of ONE source rank holding all the chunks (the others hold none): offsets = prefix sums
        mat = np.zeros((nranks, 257), dtype=np.uint32)
        mat[0, 1:] = np.cumsum(w)
        r = pkg.shard_cut_ranges(mat, nranks).astype(np.int64)
        assert r[0] == 0 and r[nranks] == 256 and (np.diff(r) >= 1).all()
        ww = w + 1                                               # (the library gives every bucket a weight of 1 on top)
        pre = np.concatenate([[0], np.cumsum(ww)])
        got = max(pre[r[i + 1]] - pre[r[i]] for i in range(nranks))
        # optimum by dynamic programming: best[k][j] = smallest possible heaviest range when the first j buckets go to k ranks
        best = np.full((nranks + 1, 257), np.iinfo(np.int64).max, dtype=np.int64)
        best[0, 0] = 0
        for k in range(1, nranks + 1):
            for j in range(k, 257):
                i = np.arange(k - 1, j)
                best[k, j] = np.minimum.reduce(np.maximum(best[k - 1, i], pre[j] - pre[i]))
        assert got == best[nranks, 256], (trial, got, best[nranks, 256])
        # round 5's greedy cut
        g = [0]
        for q in range(1, nranks):
            want = pre[256] * q // nranks
            b = g[-1] + 1
            while b < 256 - (nranks - q) and pre[b] < want:
                b += 1
            g.append(b)
        g.append(256)
        assert got <= max(pre[g[i + 1]] - pre[g[i]] for i in range(nranks))
