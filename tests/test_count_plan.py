"""CPU: the work items and launches of the count stage (csrc/sdt_count_plan.h through sdt_sk_plan_count_items) -- the host
computation between the level-2 scatter and k_sk_count.  An item is a run of the level-2 chunk list; the kernel's workgroups
take items first come first served and merge an item's nodes into the node table WITHOUT atomics when the item is flagged as holding
whole buckets only, so the plan must be airtight: every chunk in
exactly one item, a flagged item made of complete buckets, the buckets an item names exactly the buckets of its chunks -- all in one
level-1 bucket and a span of 64 --, no item across two launches.
Stands where the reference hands a batch to its threads (prlHashReads.c:312-336); the kernels it feeds are tested on the GPU."""
import numpy as np
import pytest

import __graft_entry__ as ge

PACK, ITEM, WHOLE, SPAN, L2 = 64, 2048, 0x80000000, 64, 1024        # (ITEM: SK_COUNT_ITEM_CHUNKS of csrc/sdt_pipeline.hpp, 2048 since round 6)


def _lists(rng, nb, kind):
    if kind == "tiny":                    # early, short batches: most buckets empty or a chunk or two
        n = rng.integers(0, 3, size=nb) * (rng.random(nb) < 0.4)
    elif kind == "mixed":                 # a real batch: hundreds of chunks, a few giant minimizers, some empty
        n = (rng.pareto(1.1, size=nb) * 40).astype(np.int64)
        n[rng.integers(0, nb, size=nb // 10)] = 0
        n[rng.integers(0, nb, size=3)] = rng.integers(3000, 9000, size=3)
    else:                                 # everything in one bucket
        n = np.zeros(nb, dtype=np.int64)
        n[nb // 2] = 5000
    off = np.zeros(nb + 1, dtype=np.uint32)
    off[1:] = np.cumsum(n)
    km = n * rng.integers(100, 400, size=nb)          # k-mers per bucket: 100..400 per chunk
    kp = np.zeros(nb + 1, dtype=np.uint64)
    kp[1:] = np.cumsum(km)
    return off, kp


@pytest.mark.parametrize("kind", ["tiny", "mixed", "one"])
@pytest.mark.parametrize("first_limit,limit", [(1 << 62, 1 << 62), (50_000, 400_000), (1, 1)])
def test_every_chunk_in_exactly_one_item(kind, first_limit, limit):
    pkg = ge.load_package()
    rng = np.random.default_rng(hash((kind, limit)) & 0xFFFF)
    nb = 4096
    off, kp = _lists(rng, nb, kind)
    items, first, lk = pkg.count_plan(off, kp, first_limit, limit)
    c0 = items[:, 0].astype(np.int64)
    c1 = (items[:, 1] & ~np.uint32(WHOLE)).astype(np.int64)
    whole = (items[:, 1] & np.uint32(WHOLE)) != 0
    f0, f1 = items[:, 2].astype(np.int64), items[:, 3].astype(np.int64)
    total = int(off[-1])
    # within a launch the items come largest first (by power-of-two size class, list order within a class) ...
    for li in range(len(lk)):
        a, b = int(first[li]), int(first[li + 1])
        cls = np.floor(np.log2(np.maximum(c1[a:b] - c0[a:b], 1))).astype(np.int64)
        assert (np.diff(cls) <= 0).all(), "a launch hands out its largest items first"
        for k in np.unique(cls):
            assert (np.diff(c0[a:b][cls == k]) > 0).all(), "list order within a size class"
        o = a + np.argsort(c0[a:b], kind="stable")
        c0[a:b], c1[a:b], whole[a:b], f0[a:b], f1[a:b] = c0[o], c1[o], whole[o], f0[o], f1[o]
    # ... and, put back into list order, they tile the chunk list
    if total == 0:
        assert len(items) == 0
    else:
        assert c0[0] == 0 and c1[-1] == total and (c0[1:] == c1[:-1]).all() and (c1 > c0).all()
    # bucket of every list position
    bucket_of = np.repeat(np.arange(nb), np.diff(off.astype(np.int64)))
    starts = set(off[:-1][np.diff(off.astype(np.int64)) > 0].tolist())
    ends = set(off[1:][np.diff(off.astype(np.int64)) > 0].tolist())
    for a, b, w, fa, fb in zip(c0, c1, whole, f0, f1):
        nbk = len(np.unique(bucket_of[a:b]))
        # the buckets the item names are the buckets of its chunks: k_sk_count files every node under first bucket + offset
        assert fa == bucket_of[a] and fb == bucket_of[b - 1], "an item knows its first and last final bucket"
        assert fb - fa < SPAN and fa // L2 == fb // L2, "an item stays inside one level-1 bucket and a span of 64 final buckets"
        if w:
            assert a in starts and b in ends, "a flagged item must be made of complete buckets"
            assert b - a <= ITEM
            if nbk > 1:
                assert b - a <= PACK, "only small buckets share an item"
        else:
            assert nbk == 1 and b - a <= ITEM, "a piece of a giant bucket lies inside it"
            f = bucket_of[a]
            assert int(off[f + 1]) - int(off[f]) > ITEM
    # launches: cut between buckets, k-mers add up, limits respected unless a single bucket is larger
    assert first[0] == 0 and first[-1] == len(items) and (np.diff(first.astype(np.int64)) >= 0).all()
    assert int(lk.sum()) == int(kp[-1])
    for li in range(len(lk)):
        its = range(int(first[li]), int(first[li + 1]))
        if len(its) == 0:
            continue
        fs = np.unique(bucket_of[c0[its[0]]: c1[its[-1]]])
        if li > 0 and first[li] > 0:      # no bucket and no item on both sides of a launch boundary
            prev_last = bucket_of[c1[int(first[li]) - 1] - 1]
            assert prev_last < fs[0]
        cap = first_limit if li == 0 else limit
        nonempty = [f for f in fs if kp[f + 1] > kp[f]]
        assert int(lk[li]) <= cap or len(nonempty) == 1


def test_small_neighbours_share_an_item_and_big_ones_do_not():
    pkg = ge.load_package()
    n = np.array([3, 0, 5, 0, 0, 7, 60, 2, 4000, 1, 1], dtype=np.int64)
    off = np.zeros(len(n) + 1, dtype=np.uint32)
    off[1:] = np.cumsum(n)
    kp = (off.astype(np.uint64) * np.uint64(100))
    items, first, lk = pkg.count_plan(off, kp, 1 << 62, 1 << 62)
    got = [(int(a), int(b & 0x7FFFFFFF), bool(b & WHOLE)) for a, b, _, _ in items]
    assert [(int(x), int(y)) for _, _, x, y in items] == [(8, 8), (8, 8), (6, 7), (0, 5), (9, 10)]
    # 3 + 5 + 7 = 15 chunks share the first item; 60 more would make 75 > 64: its own item, which 2 more (62) may join;
    # the 4000-chunk bucket is cut in two pieces, not flagged; the two single chunks behind it share the last item
    # (handed out largest first: 2048 and 1952 chunks, then 62, 15, 2)
    assert got == [(77, 2125, False), (2125, 4077, False), (15, 77, True), (0, 15, True), (4077, 4079, True)]
    assert len(lk) == 1 and int(lk[0]) == int(kp[-1])


def test_sparse_small_buckets_do_not_share_an_item_across_a_span_or_a_level_one_bucket():
    pkg = ge.load_package()
    n = np.zeros(3000, dtype=np.int64)
    n[[0, 63, 64, 1000, 1023, 1024, 1030, 2047, 2048]] = 1
    off = np.zeros(len(n) + 1, dtype=np.uint32)
    off[1:] = np.cumsum(n)
    kp = (off.astype(np.uint64) * np.uint64(100))
    items, first, lk = pkg.count_plan(off, kp, 1 << 62, 1 << 62)
    spans = sorted((int(x), int(y)) for _, _, x, y in items)
    # 0 and 63 share (span 64), 64 starts anew; 1000 and 1023 share, 1024 is another level-1 bucket; 1030 joins it; 2047 and 2048 part
    assert spans == [(0, 63), (64, 64), (1000, 1023), (1024, 1030), (2047, 2047), (2048, 2048)]


def test_bad_arguments():
    pkg = ge.load_package()
    off = np.zeros(3, dtype=np.uint32)
    kp = np.zeros(3, dtype=np.uint64)
    with pytest.raises(pkg.SdtError):
        pkg.count_plan(off, kp, 0, 10)
    with pytest.raises(pkg.SdtError):
        pkg.count_plan(off, kp, 10, 10, max_launches=0)
