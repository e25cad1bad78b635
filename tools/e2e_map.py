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
ap.add_argument("--layout", choices=["pe", "mixed"], default="pe")
ap.add_argument("--pregraph", choices=["ref", "ours"], default="ref",
                help="who makes the pregraph files the reference's contig reads (ours = sdt-pregraph: byte-identical output, minutes faster)")
args = ap.parse_args()

tmp = tempfile.mkdtemp(prefix="sdt_e2emap_")
try:
    tx = synth.make_transcriptome(args.T, seed=42)
    t0 = time.time()
    import numpy as np

    def write_pairs(p1, p2, n_pairs, seed0, ins, ragged=0, fasta=False, interleaved=None):
        """FASTQ pair (or FASTA pair / one interleaved FASTA); ragged > 0: every read cut to a random length >= ragged"""
        rng = np.random.default_rng(seed0)
        qual = b"I" * args.read_len
        outs = [open(interleaved, "wb")] if interleaved else [open(p1, "wb"), open(p2, "wb")]
        done = 0
        while done < n_pairs:
            n = min(250_000, n_pairs - done)
            (c1, _), (c2, _) = synth.sample_pairs(*tx, n_pairs=n, read_len=args.read_len, seed=seed0 + done, err=0.002, avg_ins=min(ins, 600))
            l1, l2 = synth.BASES[c1].reshape(n, args.read_len), synth.BASES[c2].reshape(n, args.read_len)
            len1 = rng.integers(ragged, args.read_len + 1, size=n) if ragged else np.full(n, args.read_len)
            len2 = rng.integers(ragged, args.read_len + 1, size=n) if ragged else np.full(n, args.read_len)
            if interleaved:
                outs[0].write(b"".join(b">p%d/1\n%s\n>p%d/2\n%s\n" % (done + i, l1[i, :len1[i]].tobytes(), done + i, l2[i, :len2[i]].tobytes()) for i in range(n)))
            elif fasta:
                outs[0].write(b"".join(b">r%d/1\n%s\n" % (done + i, l1[i, :len1[i]].tobytes()) for i in range(n)))
                outs[1].write(b"".join(b">r%d/2\n%s\n" % (done + i, l2[i, :len2[i]].tobytes()) for i in range(n)))
            else:
                outs[0].write(b"".join(b"@r%d/1\n%s\n+\n%s\n" % (done + i, l1[i, :len1[i]].tobytes(), qual[:len1[i]]) for i in range(n)))
                outs[1].write(b"".join(b"@r%d/2\n%s\n+\n%s\n" % (done + i, l2[i, :len2[i]].tobytes(), qual[:len2[i]]) for i in range(n)))
            done += n
        for o in outs:
            o.close()
        for f in ([interleaved] if interleaved else [p1, p2]):
            if os.path.getsize(f) % 32768 == 0:
                open(f, "ab").write(b"\n")

    cfg = os.path.join(tmp, "lib.cfg")
    pg_cfg = cfg
    if args.layout == "pe":
        f1, f2 = os.path.join(tmp, "r_1.fq"), os.path.join(tmp, "r_2.fq")
        write_pairs(f1, f2, args.pairs, 1000, 300)
        with open(cfg, "w") as fo:
            fo.write(f"max_rd_len={args.read_len}\n[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\nq1={f1}\nq2={f2}\n")
    else:
        # mixed: a 2500-bp reverse_seq library with ragged reads, a 300-bp library with two FASTQ pairs, a 500-bp library
        # (map_len=60) with an interleaved FASTA and a FASTA pair, a single-end file (ignored by map)
        q = args.pairs // 5
        a1, a2 = os.path.join(tmp, "a_1.fq"), os.path.join(tmp, "a_2.fq")
        b1, b2 = os.path.join(tmp, "b_1.fq"), os.path.join(tmp, "b_2.fq")
        c1_, c2_ = os.path.join(tmp, "c_1.fq"), os.path.join(tmp, "c_2.fq")
        pf = os.path.join(tmp, "d_p.fa")
        e1, e2 = os.path.join(tmp, "e_1.fa"), os.path.join(tmp, "e_2.fa")
        write_pairs(a1, a2, q, 1000, 2500, ragged=args.K - 3)
        write_pairs(b1, b2, q, 2000000, 300)
        write_pairs(c1_, c2_, q, 4000000, 300)
        write_pairs(None, None, q, 6000000, 500, interleaved=pf)
        write_pairs(e1, e2, args.pairs - 4 * q, 8000000, 500, fasta=True)
        with open(cfg, "w") as fo:
            fo.write(f"max_rd_len={args.read_len}\n[LIB]\navg_ins=2500\nreverse_seq=1\nasm_flags=3\nq1={a1}\nq2={a2}\n"
                     f"[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\nq1={b1}\nq2={b2}\nq1={c1_}\nq2={c2_}\nq={b1}\n"
                     f"[LIB]\navg_ins=500\nreverse_seq=0\nasm_flags=2\nmap_len=60\np={pf}\nf1={e1}\nf2={e2}\n")
        # the reference's pregraph hangs on some FASTA inputs: the graph is built from the FASTQ libraries only
        pg_cfg = os.path.join(tmp, "pg.cfg")
        with open(pg_cfg, "w") as fo:
            fo.write(f"max_rd_len={args.read_len}\n[LIB]\navg_ins=300\nreverse_seq=0\nasm_flags=3\nq1={b1}\nq2={b2}\nq1={c1_}\nq2={c2_}\n")
    res = {"pairs": args.pairs, "reads": 2 * args.pairs, "read_len": args.read_len, "K": args.K, "p": args.p,
           "kmers": 2 * args.pairs * (args.read_len - args.K + 1), "gen_s": round(time.time() - t0, 1)}
    ref = os.path.join(ROOT, "oracle", "_ref", f"SOAPdenovo-Trans-{31 if args.K <= 31 else 127}mer")
    g_ref, g_ours = os.path.join(tmp, "ref"), os.path.join(tmp, "ours")
    t0 = time.time()
    pg = [ref, "pregraph"] if args.pregraph == "ref" else [os.path.join(pkg.CSRC_DIR, "sdt-pregraph"), "pregraph"]
    subprocess.run(pg + ["-s", pg_cfg, "-K", str(args.K), "-p", str(args.p), "-o", g_ref], check=True, capture_output=True, timeout=args.timeout)
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
