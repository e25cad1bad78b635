"""repeat the hot-bucket input through the locality pipeline and print the k-mers counted and the pipeline counters each time
(N=<runs> in the environment; tools/README.md)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd.synth import pack_2bit
K = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n, L = 1500, 100
codes = np.zeros(n * L, dtype=np.uint8)
codes[L * 1000:] = np.tile(np.array([0, 1, 2, 3, 3, 1], dtype=np.uint8), (n - 1000) * L // 6 + 1)[: (n - 1000) * L]
offs = (np.arange(n + 1) * L).astype(np.uint64)
res = []
for it in range(int(os.environ.get("N", "12"))):
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=2) as g:
        g.push_reads(pack_2bit(codes), offs)
        res.append(g.finish_count())
        st = g.stage_times()[1]
        print(res[-1], {k: st[k] for k in ("chunks_l1", "chunks_l2", "pool_direct", "merges", "cnt_ticks_setup", "cnt_ticks_fill", "cnt_ticks_count", "cnt_ticks_merge", "sc_ticks_stage", "sc_ticks_minima", "sc_ticks_starts", "sc_ticks_emit")})
print(sorted(set(res)), len([r for r in res if r[0] != n * (L - K + 1)]), "bad of", len(res))
