"""Multi-GPU product path (include/sdt_gpu.h "bucket sharding"): one process per rank, C entry points only.

CPU: the shared-memory transport's control plane with 3 processes (no device).
GPU: 2 and 4 ranks sharing the one GPU of the box over the shared-memory transport (RCCL refuses two ranks per device):
the union of the ranks' tables must be the single-rank table -- keys, link counters, counts, first-occurrence
ordinals -- every node must sit on the rank that owns its bucket, and the summed kmerFreq must be the oracle's.
RCCL itself is exercised with one rank (library load, communicator, control-plane collectives)."""
import multiprocessing as mp
import os
import sys
import uuid

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _selftest_worker(name, rank, n, q):
    import __graft_entry__ as ge
    pkg = ge.load_package()
    lib = pkg.load_library()
    rc = lib.sdt_comm_selftest_shm(name.encode(), rank, n, 50)
    q.put((rank, rc, lib.sdt_gpu_last_error().decode() if rc else ""))


@pytest.mark.parametrize("n", [2, 3])
def test_shm_transport_control_plane(n):
    """all-gather / all-reduce / barriers of the exchange layer between n processes (the N>1 control path on CPU)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "t" + uuid.uuid4().hex[:12]
    ps = [ctx.Process(target=_selftest_worker, args=(name, r, n, q)) for r in range(n)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
    assert sorted(r for r, _, _ in res) == list(range(n))
    assert all(rc == 0 for _, rc, _ in res), res


def test_kmer_owner_is_strand_symmetric():
    """the owner of a k-mer is a function of its canonical minimizer: a k-mer and its reverse complement agree"""
    import __graft_entry__ as ge
    import oracle_binding as ob
    pkg = ge.load_package()
    rng = np.random.default_rng(5)
    for K in (21, 31, 47, 63, 95, 127):
        nw = 1 if K <= 31 else (2 if K <= 63 else 4)
        for _ in range(40):
            codes = rng.integers(0, 4, size=K, dtype=np.uint8)
            rc = (codes[::-1] ^ 2).astype(np.uint8)
            owners = set()
            for s in (codes, rc):
                v = 0
                for b in s:
                    v = (v << 2) | int(b)
                words = [(v >> (64 * (nw - 1 - i))) & 0xFFFFFFFFFFFFFFFF for i in range(nw)]
                owners.add(pkg.kmer_owner(words, K, 8))
            assert len(owners) == 1 and 0 <= owners.pop() < 8


def _shard_worker(name, rank, n, K, nreads, L, out, env):
    os.environ.update(env)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth
    tx = synth.make_transcriptome(30, seed=K)
    codes, offs = synth.sample_reads(*tx, n_reads=nreads, read_len=L, seed=K + 1, err=0.003, ragged=True)
    lo, hi = rank * nreads // n, (rank + 1) * nreads // n
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=pkg.SDT_FLAG_TRACK_FIRST) as g:
        g.comm_init_shm(name, rank, n)
        g.set_read_ordinal(lo, 1)
        # three collective calls of different sizes; the last rank has nothing for the second one
        cuts = [lo, lo + (hi - lo) // 3, lo + (hi - lo) // 3 if rank == n - 1 else lo + 2 * (hi - lo) // 3, hi]
        for a, b in zip(cuts[:-1], cuts[1:]):
            c = codes[int(offs[a]):int(offs[b])]
            g.push_reads_sharded(synth.pack_2bit(c) if b > a else np.zeros(4, dtype=np.uint32), offs[a:b + 1] - offs[a])
        kmers, nodes = g.finish_count()
        hist, linear = g.mark_and_hist()
        tot = g.allreduce(np.concatenate([hist, [kmers, nodes, linear]]))
        keys, l, rf, cnt, first = g.export_nodes(with_first=True)
        sent, recv, ms, nx = g.comm_stats()
        np.savez(out, keys=keys, l=l, rf=rf, cnt=cnt, first=first, tot=tot, local=np.array([kmers, nodes]), comm=np.array([sent, recv, nx]),
                 ranges=g.shard_ranges(n))


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,env", [(2, 31, {}), (4, 31, {"SDT_SHARD_ROUND_KMERS": "40000"}), (2, 63, {"SDT_SHARD_ROUND_KMERS": "60000"}),
                                     (4, 25, {"SDT_SHARD_ROUND_KMERS": "50000", "SDT_SHARD_RECV_CHUNKS": "300"})])
def test_sharded_ranks_sharing_one_gpu(tmp_path, n, K, env):
    import __graft_entry__ as ge
    import oracle_binding as ob
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth
    nreads, L = 6000, 120
    ctx = mp.get_context("spawn")
    name = "g" + uuid.uuid4().hex[:12]
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(n)]
    ps = [ctx.Process(target=_shard_worker, args=(name, r, n, K, nreads, L, outs[r], env)) for r in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
        assert p.exitcode == 0
    # the same reads through the oracle, and their first-occurrence ordinals
    tx = synth.make_transcriptome(30, seed=K)
    codes, offs = synth.sample_reads(*tx, n_reads=nreads, read_len=L, seed=K + 1, err=0.003, ragged=True)
    o = ob.Oracle(K, nsets=4)
    o.add_reads(codes, offs)
    ohist, olinear = o.mark()
    okeys, ol, orr, ocnt, ofl = o.export()
    ofirst = o.export_first()
    nw = okeys.shape[1]
    kw = ob.key_words_for(K)

    def kint(row):
        v = 0
        for x in row:
            v = (v << 64) | int(x)
        return v
    want = {kint(k): (int(a), int(b), int(c), int(f)) for k, a, b, c, f in zip(okeys, ol, orr, ocnt, ofirst)}
    got = {}
    ranges = np.load(outs[0])["ranges"]
    assert ranges[0] == 0 and ranges[n] == 256 and (np.diff(ranges.astype(np.int64)) >= 1).all()
    sizes = []
    for r in range(n):
        z = np.load(outs[r])
        assert (z["ranges"] == ranges).all(), "the ranks disagree about who owns what"
        sizes.append(int(z["local"][0]))
        assert int(z["tot"][257]) == o.kmers_in_reads() and int(z["tot"][258]) == o.node_count() and int(z["tot"][259]) == olinear
        assert (z["tot"][:257] == ohist).all()
        assert len(z["keys"]) == int(z["local"][1])
        for k, a, b, c, f in zip(z["keys"], z["l"], z["rf"], z["cnt"], z["first"]):
            ki = kint(k)
            assert ki not in got, "a k-mer sits on two ranks"
            got[ki] = (int(a), int(b) & 0xFFFFFF, int(c), int(f))
            assert ranges[r] <= pkg.kmer_bucket([int(x) for x in k], K) < ranges[r + 1]
        if n > 1:
            assert int(z["comm"][0]) > 0 and int(z["comm"][2]) >= 2
    assert got == want
    # the ranges were cut by weight: no rank counts more than 1.6x the mean (equal ranges of these buckets: up to 2x)
    assert max(sizes) <= 1.6 * sum(sizes) / n, sizes


def _overflow_worker(name, rank, n, K, nreads, L, q):
    os.environ["SDT_SK_POOL_CHUNKS1"] = "48"
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from soapdenovo_trans_amd import synth
    tx = synth.make_transcriptome(30, seed=K)
    codes, offs = synth.sample_reads(*tx, n_reads=nreads, read_len=L, seed=K + 1, err=0.003)
    lo, hi = rank * nreads // n, (rank + 1) * nreads // n
    try:
        with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=pkg.SDT_FLAG_TRACK_FIRST) as g:
            g.comm_init_shm(name, rank, n)
            g.set_read_ordinal(lo, 1)
            c = codes[int(offs[lo]):int(offs[hi])]
            g.push_reads_sharded(synth.pack_2bit(c), offs[lo:hi + 1] - offs[lo])
            g.finish_count()
        q.put((rank, "no error"))
    except pkg.SdtError as e:
        q.put((rank, int(e.code)))


@pytest.mark.gpu
def test_sharded_pool_overflow_is_an_error_not_a_local_insert(pkg):
    """a rank whose level-1 pool overflows must NOT fall back on its local table (the key may belong to another rank: the same
    k-mer would become a node twice): with a pool of 48 chunks (test hook) every rank reports SDT_EFULL"""
    n, K, nreads, L = 2, 31, 6000, 120
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "o" + uuid.uuid4().hex[:12]
    ps = [ctx.Process(target=_overflow_worker, args=(name, r, n, K, nreads, L, q)) for r in range(n)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=400) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(r, pkg.SDT_EFULL) for r in range(n)], res


@pytest.mark.gpu
def test_rccl_single_rank_control_plane(pkg, synth):
    """one rank: librccl is loaded, a communicator made, the control-plane collectives run through ncclAllGather"""
    cid = pkg.new_comm_id()
    assert len(cid) == 128 and any(cid)
    with pkg.PregraphGPU(31, est_distinct=1 << 16) as g:
        g.comm_init(cid, 0, 1)
        assert (g.allreduce([3, -4, 1 << 40]) == np.array([3, -4, 1 << 40])).all()
        tx = synth.make_transcriptome(10, seed=3)
        codes, offs = synth.sample_reads(*tx, n_reads=2000, read_len=100, seed=4)
        g.push_reads_sharded(synth.pack_2bit(codes), offs)
        kmers, nodes = g.finish_count()
        assert kmers == 2000 * (100 - 31 + 1) and nodes > 0
