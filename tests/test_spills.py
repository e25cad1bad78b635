"""GPU (-m gpu): the count stage's SPILL path -- a k-mer that finds no slot in the workgroup's LDS table goes to the node table with a
memory-side atomic, on a table the same workgroup also writes with plain stores (the owned flush).  The product's LDS table has 1280..2048
slots and spills a few thousand k-mers per 24 G; here the library is the test build `libsdt_gpu_smalllds.so` (csrc/Makefile:
-DSDT_SK_TEST_SLOTS=64, everything else the same code), whose table overflows all the time: flush, guarded round, spilling rounds,
non-owned flush, one after the other within every work item.  Every node is compared with the oracle (newhash.c:71-96 semantics:
count, 8 saturating link counters, flags).

What is pinned: the ordering rule of k_sk_count -- no memory-side atomic before every wave's plain stores of the last owned flush
have landed (the round behind a flush cannot spill: `guard`; at the barrier that ends it every wave has waited for its stores).  A
lost store or a lost atomic shows as a wrong count or a wrong link counter of some node.

The variant library is loaded in a child process (SDT_GPU_LIB): this process has the product library loaded."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import synth
import oracle_binding as ob
from test_gpu_parity import node_dict_gpu, node_dict_oracle

K, L, track, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pkg.load_library()
assert pkg.LIB_PATH.endswith("libsdt_gpu_smalllds.so"), pkg.LIB_PATH
# a few transcripts at deep coverage (the same keys come back generation after generation: merges into nodes that plain stores wrote a
# moment ago) + a hot repeat (one bucket, a handful of keys, thousands of occurrences) + noise (new keys all the time: the table overflows)
tx = synth.make_transcriptome(6, seed=K)
codes, offs = synth.sample_reads(*tx, n_reads=9000, read_len=L, seed=K + 1, err=0.01)
hot = np.tile(np.array([0, 1, 2, 3, 3, 1], dtype=np.uint8), 1500 * L // 6 + 1)[: 1500 * L]
rng = np.random.default_rng(K)
noise = rng.integers(0, 4, size=1500 * L, dtype=np.uint8)
codes = np.concatenate([codes, hot, noise])
offs = (np.arange(len(codes) // L + 1, dtype=np.uint64) * L)
words = synth.pack_2bit(codes)
o = ob.Oracle(K, nsets=4)
o.add_reads(codes, offs)
ohist, olinear = o.mark()
want = node_dict_oracle(o)
spills = merges = 0
flags = pkg.SDT_FLAG_PARTITION | (pkg.SDT_FLAG_TRACK_FIRST if track else 0)
for _ in range(reps):
    with pkg.PregraphGPU(K, est_distinct=1 << 18, flags=flags) as g:
        g.push_reads(words, offs)
        kmers, nodes = g.finish_count()
        assert (kmers, nodes) == (o.kmers_in_reads(), o.node_count()), (kmers, nodes, o.kmers_in_reads(), o.node_count())
        hist, linear = g.mark_and_hist()
        assert linear == olinear and (hist == ohist).all()
        got = node_dict_gpu(g)
        assert got == want, [(hex(k), got.get(k), want.get(k)) for k in want if got.get(k) != want.get(k)][:5]
        c = g.stage_times()[1]
        spills += c["lds_spills"]; merges += c["merges"]
print(json.dumps({"lds_spills": spills, "merges": merges, "nodes": nodes, "kmers": kmers}))
"""


@pytest.mark.parametrize("K,L,track", [(21, 100, 0), (31, 150, 0), (31, 150, 1), (45, 150, 0), (75, 200, 0)])
def test_spilling_count_stage_equals_oracle(pkg, K, L, track):
    lib = os.path.join(pkg.CSRC_DIR, "libsdt_gpu_smalllds.so")
    assert os.path.exists(lib), "csrc/Makefile builds it beside libsdt_gpu.so (__graft_entry__.build)"
    env = dict(os.environ, SDT_GPU_LIB=lib)
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD, str(K), str(L), str(track), "20"], capture_output=True, text=True,
                       env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    # the path under test ran: thousands of k-mers went to the node table by memory-side atomics between owned flushes
    assert res["lds_spills"] > 1000, res
