#!/usr/bin/env python3
"""CPU-only: time the host graph phases (sdt-graphcheck) on a node table built by the oracle.
usage: bench_host_graph.py [reads] [p]"""
import os, struct, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import synth
import oracle_binding as ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K, L = 31, 150
tx = synth.make_transcriptome(max(20, n // 1000), seed=42)
codes, offs = synth.sample_reads(*tx, n_reads=n, read_len=L, seed=7, err=0.002)
t0 = time.time()
o = ob.Oracle(K, nsets=3)
o.add_reads(codes, offs)
o.mark()
keys, l, r, cnt, fl = o.export()
fo = o.export_first()
print(f"oracle: {len(keys)} nodes in {time.time() - t0:.1f} s", file=sys.stderr)
rflags = (r.astype(np.uint32) | ((fl & 1).astype(np.uint32) << 24) | (((fl >> 1) & 1).astype(np.uint32) << 25)
          | (((fl >> 2) & 1).astype(np.uint32) << 27))
dump = "/tmp/sdt_nodes.bin"
with open(dump, "wb") as f:
    f.write(struct.pack("<6iQ", K, 1, 1, p, 0, 5, len(keys)))
    f.write(np.ascontiguousarray(keys[:, 3:]).tobytes())
    f.write(l.astype(np.uint32).tobytes()); f.write(rflags.tobytes()); f.write(cnt.astype(np.uint32).tobytes()); f.write(fo.tobytes())
r = subprocess.run([os.path.join(pkg.CSRC_DIR, "sdt-graphcheck"), dump, "/tmp/sdt_out"], capture_output=True, text=True)
print(r.stderr[-2000:])
print("\n".join(x for x in r.stdout.splitlines() if "off" in x or "edges" in x or "vertex" in x))
