"""GPU (-m gpu): the node log and the bucket-major node table of the locality pipeline (csrc/sdt_bm_kernels.cuh) against the
oracle's node table (put_kmerset / update_kmer, newhash.c:71-96,411-462: count, eight saturating link counters, `single`).
What pass 1 leaves is a pure function of the multiset of (k-mer, prev, next) occurrences, so every way of getting there -- one
fold, a fold over an earlier table, flat nodes folded in, buckets cut into sub-buckets, buckets merged in several passes -- must
end in the oracle's table, bit for bit; and every node must be found again by key (the look-ups compute the key's minimizer
bucket: sdt_table.cuh probe_begin)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import node_dict_gpu, node_dict_oracle, keys_to_int

pytestmark = pytest.mark.gpu

PIPE = 2 | 64     # SDT_FLAG_PARTITION | SDT_FLAG_NODE_LOG (multi-word keys take the flat merges by default)


@pytest.fixture
def knobs():
    """test hooks of the fold (sdt_gpu.hip bm_fold): read from the environment at every fold"""
    names = ("SDT_BM_GIANT", "SDT_BM_SUB_TARGET", "SDT_BM_LDS_CAP", "SDT_LOG_SLAB_LOG2")
    old = {n: os.environ.get(n) for n in names}

    def set_(**kw):
        for k, v in kw.items():
            os.environ[k] = str(v)
    yield set_
    for n, v in old.items():
        if v is None:
            os.environ.pop(n, None)
        else:
            os.environ[n] = v


def _workload(synth, K, L, n=5000, seed=0):
    tx = synth.make_transcriptome(20, seed=K + seed)
    return synth.sample_reads(*tx, n_reads=n, read_len=L, seed=K + 7 + seed, err=0.004, ragged=True)


def _check_all(pkg, g, o, with_first=False):
    assert g.finish_count() == (o.kmers_in_reads(), o.node_count())
    hist, linear = g.mark_and_hist()
    ohist, olinear = o.mark()
    assert linear == olinear and (hist == ohist).all()
    assert node_dict_gpu(g) == node_dict_oracle(o)
    info = g.table_info()
    assert info["layout"] == "bucket-major" and info["nodes"] == o.node_count()
    # every node is found again by key (k_set_index: find_slot -> table_find -> probe_begin)
    keys = g.export_nodes()[0]
    g.set_node_index(keys)
    return info


@pytest.mark.parametrize("K,L", [(21, 100), (31, 150), (47, 150), (63, 250), (95, 250), (127, 250)])
def test_a_fold_over_an_earlier_table_equals_one_fold(pkg, synth, K, L):
    """finish_count in the middle of the stream: the second fold merges the first one's table with the new segments"""
    codes, offs = _workload(synth, K, L)
    o = ob.Oracle(K, nsets=3)
    o.add_reads(codes, offs)
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=PIPE) as g:
        cut = [0, len(offs) // 3, 2 * len(offs) // 3, len(offs) - 1]
        for a, b in zip(cut[:-1], cut[1:]):
            part = codes[int(offs[a]): int(offs[b])]
            g.push_reads(synth.pack_2bit(part), offs[a: b + 1] - offs[a])
            g.finish_count()                                   # a fold per push
        info = _check_all(pkg, g, o)
        assert info["folds"] == 3


@pytest.mark.parametrize("K,L", [(31, 150), (63, 250), (95, 250)])
def test_first_occurrence_ordinals_through_folds(pkg, synth, K, L):
    """SDT_FLAG_TRACK_FIRST: the smallest (read << 16 | position) of a key's occurrences survives segments, folds and sub-buckets"""
    codes, offs = _workload(synth, K, L, n=4000, seed=3)
    o = ob.Oracle(K, nsets=3)
    o.add_reads(codes, offs)
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=PIPE | pkg.SDT_FLAG_TRACK_FIRST) as g:
        half = len(offs) // 2
        g.push_reads(synth.pack_2bit(codes[: int(offs[half])]), offs[: half + 1])
        g.finish_count()
        g.push_reads(synth.pack_2bit(codes[int(offs[half]):]), offs[half:] - offs[half])
        assert g.finish_count() == (o.kmers_in_reads(), o.node_count())
        keys, _, _, _, first = g.export_nodes(with_first=True)
        got = dict(zip(keys_to_int(keys), (int(x) for x in first)))
    # the oracle of first occurrences: a direct-family context (one atomic min per occurrence)
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=1 | pkg.SDT_FLAG_TRACK_FIRST) as g:
        g.push_reads(synth.pack_2bit(codes), offs)
        g.finish_count()
        keys, _, _, _, first = g.export_nodes(with_first=True)
        want = dict(zip(keys_to_int(keys), (int(x) for x in first)))
    assert got == want


@pytest.mark.parametrize("K,L", [(23, 100), (31, 150), (63, 250), (127, 250)])
@pytest.mark.parametrize("giant,sub,cap", [(64, 32, 0), (1 << 30, 1 << 30, 24), (96, 16, 40)])
def test_sub_buckets_and_several_parts(pkg, synth, knobs, K, L, giant, sub, cap):
    """giant buckets are cut into sub-buckets merged side by side, buckets past the LDS table are merged in several passes: forced
    on a small input by shrinking the thresholds (sub-buckets only / parts only / both)"""
    knobs(SDT_BM_GIANT=giant, SDT_BM_SUB_TARGET=sub, SDT_BM_LDS_CAP=cap)
    codes, offs = _workload(synth, K, L, n=6000, seed=5)
    o = ob.Oracle(K, nsets=3)
    o.add_reads(codes, offs)
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=PIPE) as g:
        half = len(offs) // 2
        g.push_reads(synth.pack_2bit(codes[: int(offs[half])]), offs[: half + 1])
        g.finish_count()                                       # (the second fold reads sub-bucket ranges of the first)
        g.push_reads(synth.pack_2bit(codes[int(offs[half]):]), offs[half:] - offs[half])
        info = _check_all(pkg, g, o)
        if cap:
            assert info["max_parts"] > 1, "the small LDS cap must force several parts"


def test_many_small_slabs_of_the_node_log(pkg, synth, knobs):
    """the log grows slab by slab (the hard bound of a launch never fits a 64 K-entry slab twice): segments of all slabs in one fold"""
    knobs(SDT_LOG_SLAB_LOG2=16)
    K, L = 31, 150
    codes, offs = _workload(synth, K, L, n=20000, seed=9)
    o = ob.Oracle(K, nsets=3)
    o.add_reads(codes, offs)
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=PIPE) as g:
        step = len(offs) // 5
        for a in range(0, len(offs) - 1, step):
            b = min(a + step, len(offs) - 1)
            g.push_reads(synth.pack_2bit(codes[int(offs[a]): int(offs[b])]), offs[a: b + 1] - offs[a])
        _check_all(pkg, g, o)


def test_flat_nodes_are_folded_in(pkg, synth):
    """a context that counted a first batch with the direct family (flat table) and a second one through the pipeline: one table"""
    K, L = 31, 150
    codes, offs = _workload(synth, K, L, n=6000, seed=11)
    o = ob.Oracle(K, nsets=3)
    o.add_reads(codes, offs)
    os.environ["SDT_SK_POOL_CHUNKS1"] = "40"                    # a pool of 40 chunks: most records find none and go to the flat table
    try:
        with pkg.PregraphGPU(K, est_distinct=1 << 18, flags=PIPE) as g:
            g.push_reads(synth.pack_2bit(codes), offs)
            info = _check_all(pkg, g, o)
            assert info["folds"] >= 1
    finally:
        os.environ.pop("SDT_SK_POOL_CHUNKS1", None)
