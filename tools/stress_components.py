#!/usr/bin/env python3
"""CPU only: the component-parallel commits of the device path (sdt-graphcheck with SDT_GRAPHCHECK_EMULATE=1: labelled walks /
junction records made by the host) against the host path's sequential sweeps, on random graphs with dense junctions
(few transcripts, deep coverage, high error rates): every output file and every counter line must be identical.
usage: stress_components.py [rounds] [reads]"""
import os, struct, subprocess, sys, filecmp, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import synth
import oracle_binding as ob

def one(seed, n, K, L, ntx, err, p, d=0, dd=5):
    tx = synth.make_transcriptome(ntx, seed=seed)
    codes, offs = synth.sample_reads(*tx, n_reads=n, read_len=L, seed=seed + 1, err=err)
    nwk = ob.key_words_for(K)
    o = ob.Oracle(K, nsets=3, nw=nwk)
    o.add_reads(codes, offs)
    if d: o.delow(d)
    o.mark()
    keys, l, r, cnt, fl = o.export()
    fo = o.export_first()
    rflags = (r.astype(np.uint32) | ((fl & 1).astype(np.uint32) << 24) | (((fl >> 1) & 1).astype(np.uint32) << 25)
              | (((fl >> 2) & 1).astype(np.uint32) << 27))
    dump = "/tmp/sdt_stress_nodes.bin"
    nwv = 1 if K <= 31 else (2 if K <= 63 else 4)
    with open(dump, "wb") as f:
        f.write(struct.pack("<6iQ", K, nwv, nwk, p, d, dd, len(keys)))
        f.write(np.ascontiguousarray(keys[:, 4 - nwk:]).tobytes())
        f.write(l.astype(np.uint32).tobytes()); f.write(rflags.tobytes()); f.write(cnt.astype(np.uint32).tobytes()); f.write(fo.tobytes())
    exe = os.path.join(pkg.CSRC_DIR, "sdt-graphcheck")
    outs = {}
    for mode in ("host", "emu"):
        env = dict(os.environ, SDT_TIMING="1")
        if mode == "emu": env["SDT_GRAPHCHECK_EMULATE"] = "1"
        r = subprocess.run([exe, dump, f"/tmp/sdt_stress_{mode}"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = r
    comp = [x.strip() for x in outs["emu"].stderr.splitlines() if "components" in x]
    assert outs["host"].stdout == outs["emu"].stdout, (seed, "stdout differs")
    for ext in ("vertex", "preGraphBasic"):
        assert filecmp.cmp(f"/tmp/sdt_stress_host.{ext}", f"/tmp/sdt_stress_emu.{ext}", shallow=False), (seed, ext)
    assert gzip.open("/tmp/sdt_stress_host.edge.gz").read() == gzip.open("/tmp/sdt_stress_emu.edge.gz").read(), (seed, "edge")
    tips = [x for x in outs["emu"].stdout.splitlines() if "off" in x]
    print(f"seed {seed}: n={n} K={K} L={L} tx={ntx} err={err} p={p} d={d}: {len(keys)} nodes; {tips}; {comp}", flush=True)

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    rng = np.random.default_rng(2024)
    for i in range(rounds):
        K = int(rng.choice([21, 25, 31, 31, 41, 63]))
        L = int(rng.choice([100, 150]))
        if L <= K + 5: L = 150
        one(seed=1000 + i, n=n, K=K, L=L, ntx=int(rng.choice([3, 8, 20, 60])), err=float(rng.choice([0.005, 0.01, 0.02, 0.04])),
            p=int(rng.choice([1, 2, 3, 8, 16])), d=int(rng.choice([0, 0, 1])))
    print("all identical")
