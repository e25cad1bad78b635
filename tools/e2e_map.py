#!/usr/bin/env python3
"""End-to-end wall clock of the `map` stage: reference `map` vs `sdt-map` on the same paired FASTQ and the same
contigs (made by the reference's pregraph + contig), all output files compared byte for byte.  GPU box:
    python tools/e2e_map.py --pairs 1000000 --read-len 150 --K 31 --p 16
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from soapdenovo_trans_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1_000_000)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--K", type=int, default=31)
ap.add_argument("--p", type=int, default=16)
ap.add_argument("--T", type=int, default=2000)
ap.add_argument("--timeout", type=int, default=600)
ap.add_argument("--pregraph", choices=["ref", "ours"], default="ref",
                help="who makes the pregraph files the reference's contig reads (ours = sdt-pregraph: byte-identical output, minutes faster)")
args = ap.parse_args()

tmp = tempfile.mkdtemp(prefix="sdt_e2emap_")
try:
    tx = synth.make_transcriptome(args.T, seed=42)
    f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
    t0 = time.time()
    qual = b"I" * args.read_len
    with open(f1, "wb") as o1, open(f2, "wb") as o2:
        done, chunk = 0, 250_000
        while done < args.pairs:
            n = min(chunk, args.pairs - done)
            (c1, _), (c2, _) = synth.sample_pairs(*tx, n_pairs=n, read_len=args.read_len, seed=1000 + done, err=0.002, avg_ins=300)
            l1 = synth.BASES[c1].reshape(n, args.read_len)
            l2 = synth.BASES[c2].reshape(n, args.read_len)
            o1.write(b"".join(b"@r%d/1\n%s\n+\n%s\n" % (done + i, l1[i].tobytes(), qual) for i in range(n)))
            o2.write(b"".join(b"@r%d/2\n%s\n+\n%s\n" % (done + i, l2[i].tobytes(), qual) for i in range(n)))
            done += n
    for f in (f1, f2):
        if os.path.getsize(f) % 32768 == 0:
            open(f, "ab").write(b"\n")
    cfg = os.path.join(tmp, "lib.cfg")
    with open(cfg, "w") as fo:
        fo.write(f"max_rd_len={args.read_len}\n[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\nq1={f1}\nq2={f2}\n")
    res = {"pairs": args.pairs, "reads": 2 * args.pairs, "read_len": args.read_len, "K": args.K, "p": args.p,
           "kmers": 2 * args.pairs * (args.read_len - args.K + 1), "gen_s": round(time.time() - t0, 1)}
    ref = os.path.join(ROOT, "oracle", "_ref", f"SOAPdenovo-Trans-{31 if args.K <= 31 else 127}mer")
    g_ref, g_ours = os.path.join(tmp, "ref"), os.path.join(tmp, "ours")
    t0 = time.time()
    pg = [ref, "pregraph"] if args.pregraph == "ref" else [os.path.join(pkg.CSRC_DIR, "sdt-pregraph"), "pregraph"]
    subprocess.run(pg + ["-s", cfg, "-K", str(args.K), "-p", str(args.p), "-o", g_ref], check=True, capture_output=True, timeout=args.timeout)
    subprocess.run([ref, "contig", "-g", g_ref], check=True, capture_output=True, timeout=args.timeout)
    res["pregraph_by"] = args.pregraph
    res["pregraph_contig_s"] = round(time.time() - t0, 1)
    for ext in ("contig", "ContigIndex", "preGraphBasic"):
        shutil.copy(g_ref + "." + ext, g_ours + "." + ext)
    res["contig_bytes"] = os.path.getsize(g_ref + ".contig")
    t0 = time.time()
    rr = subprocess.run([ref, "map", "-s", cfg, "-g", g_ref, "-p", str(args.p)], capture_output=True, text=True, timeout=args.timeout)
    res["ref_map_wall_s"] = round(time.time() - t0, 2)
    res["ref_lines"] = [l for l in rr.stdout.splitlines() if "time spent" in l or "mapped to contigs" in l or "nodes allocated" in l]
    os.environ["SDT_TIMING"] = "1"
    ours = os.path.join(pkg.CSRC_DIR, "sdt-map")
    t0 = time.time()
    r = subprocess.run([ours, "map", "-s", cfg, "-g", g_ours, "-p", str(args.p)], capture_output=True, text=True, timeout=args.timeout)
    res["ours_map_wall_s"] = round(time.time() - t0, 2)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit("sdt-map failed")
    res["ours_lines"] = [l for l in r.stdout.splitlines() if "mapped to contigs" in l or "nodes allocated" in l]
    res["ours_phase_ms"] = [l.replace("[sdt-map] ", "") for l in r.stderr.splitlines() if l.startswith("[sdt-map]")]
    res["identical"] = {ext: open(g_ref + "." + ext, "rb").read() == open(g_ours + "." + ext, "rb").read()
                        for ext in ("readOnContig", "ctg2Read", "readInGap", "peGrads")}
    res["speedup_map"] = round(res["ref_map_wall_s"] / res["ours_map_wall_s"], 2)
    print(json.dumps(res, indent=1))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
