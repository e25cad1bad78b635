"""CPU, world_size 2, gloo: the N>1 path's exchange layer (soapdenovo-trans_amd/sharding.py).

The two kernels either side of the exchange need a GPU (covered by test_gpu_parity.py's virtual-rank test);
here the oracle stands in for them -- as the checker's stand-in only -- so that the ownership function, the
send-slice layout, the all-to-all(v) and the histogram all-reduce are exercised with real processes:
2 ranks x half the reads must give the single-process kmerFreq byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, n_reads, L, out_q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth, sharding
    lib = pkg.load_library()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx = synth.make_transcriptome(20, seed=3)
        codes, offs = synth.sample_reads(*tx, n_reads=n_reads, read_len=L, seed=4, ragged=True)
        lo, hi = rank * n_reads // world, (rank + 1) * n_reads // world      # this rank's slice of the reads
        nw = ob.key_words_for(K)
        rec_words = nw + 1
        keys_all, meta_all, owner_all = [], [], []
        for r in range(lo, hi):
            keys, pv, nx, _ = ob.chop_read(codes[int(offs[r]):int(offs[r + 1])], K)
            for j in range(len(keys)):
                kw = keys[j][4 - nw:]
                keys_all.append(kw)
                meta_all.append(int(pv[j]) | (int(nx[j]) << 3))
                owner_all.append(sharding.owner_of(lib, kw, world))
        owner_all = np.asarray(owner_all)
        cap = len(keys_all) + 8                                               # fixed-capacity slices
        send = torch.zeros(cap * world * rec_words, dtype=torch.int64)
        counts = torch.zeros(world, dtype=torch.int64)
        sv = send.numpy().view(np.uint64).reshape(world, cap, rec_words)
        for dst in range(world):
            idx = np.nonzero(owner_all == dst)[0]
            counts[dst] = len(idx)
            for t, i in enumerate(idx):
                sv[dst, t, :nw] = keys_all[i]
                sv[dst, t, nw] = meta_all[i]
        recv = torch.zeros(cap * world * rec_words * 2, dtype=torch.int64)
        total, rc = sharding.exchange_records(send, counts, cap, rec_words, recv)
        assert total == sum(rc)
        rv = recv.numpy().view(np.uint64)[: total * rec_words].reshape(total, rec_words)
        # every received record is ours
        for row in rv[:: max(1, total // 200)]:
            assert sharding.owner_of(lib, row[:nw], world) == rank
        o = ob.Oracle(K, nsets=1)
        L_ = ob.lib()
        sets = ob.C.cast(o.h, ob.C.POINTER(ob.SetsStruct)).contents
        set0 = ob.C.cast(sets.sets, ob.C.POINTER(ob.C.c_void_p))[0]
        for row in rv:
            w4 = [0] * (4 - nw) + [int(x) for x in row[:nw]]
            L_.sdto_set_put(set0, ob.Kmer.of(w4), int(row[nw]) & 7, (int(row[nw]) >> 3) & 7, nw, None)
        hist, linear = o.mark()
        h, k, n, l = sharding.allreduce_stats(hist, total, o.node_count(), linear, torch.device("cpu"))
        if rank == 0:
            out_q.put((h.tolist(), k, n, l))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("K", [21, 41])
def test_two_rank_exchange_matches_single_process(K):
    world, n_reads, L = 2, 300, 90
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, n_reads, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import __graft_entry__ as ge
    ge.load_package()
    from soapdenovo_trans_amd import synth
    tx = synth.make_transcriptome(20, seed=3)
    codes, offs = synth.sample_reads(*tx, n_reads=n_reads, read_len=L, seed=4, ragged=True)
    o = ob.Oracle(K, nsets=8)
    o.add_reads(codes, offs)
    hist, linear = o.mark()
    h, k, n, l = res
    assert k == o.kmers_in_reads() and n == o.node_count() and l == linear
    assert ob.kmerfreq_text(np.asarray(h)) == ob.kmerfreq_text(hist)
