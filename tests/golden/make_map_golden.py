#!/usr/bin/env python3
"""Golden fixtures for the `map` stage (prlContig2nodes prlHashCtg.c:287-425, prlRead2Ctg
prlRead2Ctg.c:656-894), made by RUNNING THE REFERENCE in this container:

    pregraph -> contig -> map      (oracle/_ref/SOAPdenovo-Trans-<variant>mer)

Per case under tests/golden/map_cases/<name>/:
  inputs   reads (gzip), lib.cfg.template, and what `map` reads of the graph: out.contig, out.ContigIndex,
           out.preGraphBasic (written by the reference's pregraph / contig for these reads)
  outputs  out.readOnContig, out.ctg2Read, out.peGrads, out.readInGap (binary), [out.readInformation with -r],
           the stdout lines that carry counters.
Run from the repo root after `make -C oracle ref`.  Fixtures are data; this script is the committed recipe.
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from soapdenovo_trans_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def write_fasta(path, codes, offs, prefix="s", width=70):
    o = offs.astype(np.int64)
    letters = synth.BASES[codes].tobytes()
    with open(path, "wb") as fo:
        for i in range(len(o) - 1):
            s = letters[o[i]:o[i + 1]]
            fo.write(b">%s%d\n" % (prefix.encode(), i))
            for j in range(0, len(s), width):          # multi-line records: the FASTA reader concatenates
                fo.write(s[j:j + width] + b"\n")


def write_fasta_interleaved(path, c1, o1, c2, o2):
    l1, l2 = synth.BASES[c1].tobytes(), synth.BASES[c2].tobytes()
    a, b = o1.astype(np.int64), o2.astype(np.int64)
    with open(path, "wb") as fo:
        for i in range(len(a) - 1):
            fo.write(b">p%d/1\n%s\n>p%d/2\n%s\n" % (i, l1[a[i]:a[i + 1]], i, l2[b[i]:b[i + 1]]))


def ragged(codes, offs, rng, lo):
    """cut every read to a random length in [lo, len]"""
    o = offs.astype(np.int64)
    keep, no = [], [0]
    for i in range(len(o) - 1):
        n = int(rng.integers(lo, o[i + 1] - o[i] + 1))
        keep.append(codes[o[i]:o[i] + n])
        no.append(no[-1] + n)
    return np.concatenate(keep), np.array(no, dtype=np.uint64)


# name, variant, K, p, libs: list of (kind, n_pairs, L, avg_ins, extra config lines, ragged_lo)
CASES = [
    dict(name="map_pe150_k31_p8", variant=31, K=31, p=8, T=30,
         libs=[dict(kind="q", n=1500, L=150, ins=200, extra="")]),
    dict(name="map_fa100_k23_p4_two_libs", variant=31, K=23, p=4, T=25,
         libs=[dict(kind="f", n=900, L=100, ins=300, extra=""),
               dict(kind="p", n=700, L=100, ins=500, extra="map_len=40\n")]),
    dict(name="map_pe250_k63_127mer_p3", variant=127, K=63, p=3, T=20,
         libs=[dict(kind="q", n=800, L=250, ins=400, extra="")]),
    dict(name="map_longins_ragged_k31_p5", variant=31, K=31, p=5, T=25, trace=True,
         libs=[dict(kind="q", n=700, L=120, ins=2500, extra="reverse_seq=1\n", ragged=20),
               dict(kind="q", n=600, L=120, ins=180, extra="", ragged=28)]),
    dict(name="map_k47_63mer_p2", variant=63, K=47, p=2, T=20,
         libs=[dict(kind="q", n=900, L=150, ins=250, extra="")]),
    # -f: the gap-filling dumps (shortreadInGap.gz, PEreadOnContig.gz); the 2500 library is filtered out of them (:439,:505)
    dict(name="map_fill_two_libs_k31_p3", variant=31, K=31, p=3, T=25, fill=True,
         libs=[dict(kind="q", n=600, L=100, ins=2500, extra="", ragged=25),
               dict(kind="q", n=800, L=100, ins=220, extra="", ragged=40)]),
]


def make_case(c):
    cdir = os.path.join(HERE, "map_cases", c["name"])
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    tx = synth.make_transcriptome(c["T"], seed=300 + len(c["name"]))
    tmp = tempfile.mkdtemp(prefix="sdtmapgold_")
    rng = np.random.default_rng(11)
    cfg = f"max_rd_len={max(l['L'] for l in c['libs'])}\n"
    pg_cfg = cfg            # pregraph + contig always get FASTQ (the reference's pregraph hangs on some FASTA inputs)
    files = []
    for li, lib in enumerate(c["libs"]):
        (c1, o1), (c2, o2) = synth.sample_pairs(*tx, n_pairs=lib["n"], read_len=lib["L"], seed=7 + li, err=0.004,
                                                avg_ins=max(lib["ins"], lib["L"]) if lib["ins"] < 1000 else 400)
        if lib.get("ragged"):
            c1, o1 = ragged(c1, o1, rng, lib["ragged"])
            c2, o2 = ragged(c2, o2, rng, lib["ragged"])
        cfg += f"[LIB]\navg_ins={lib['ins']}\nasm_flags=3\n" + (lib["extra"] if "reverse_seq" in lib["extra"] else "reverse_seq=0\n" + lib["extra"])
        g1, g2 = os.path.join(tmp, f"pg{li}_1.fq"), os.path.join(tmp, f"pg{li}_2.fq")
        synth.write_fastq(g1, c1, o1)
        synth.write_fastq(g2, c2, o2)
        pg_cfg += f"[LIB]\navg_ins={lib['ins']}\nasm_flags=3\nreverse_seq=0\nq1={g1}\nq2={g2}\n"
        if lib["kind"] == "q":
            p1, p2 = os.path.join(tmp, f"lib{li}_1.fq"), os.path.join(tmp, f"lib{li}_2.fq")
            synth.write_fastq(p1, c1, o1)
            synth.write_fastq(p2, c2, o2)
            cfg += f"q1=@DIR@/lib{li}_1.fq\nq2=@DIR@/lib{li}_2.fq\n"
            files += [p1, p2]
        elif lib["kind"] == "f":
            p1, p2 = os.path.join(tmp, f"lib{li}_1.fa"), os.path.join(tmp, f"lib{li}_2.fa")
            write_fasta(p1, c1, o1)
            write_fasta(p2, c2, o2)
            cfg += f"f1=@DIR@/lib{li}_1.fa\nf2=@DIR@/lib{li}_2.fa\n"
            files += [p1, p2]
        else:
            p1 = os.path.join(tmp, f"lib{li}_p.fa")
            write_fasta_interleaved(p1, c1, o1, c2, o2)
            cfg += f"p=@DIR@/lib{li}_p.fa\n"
            files += [p1]
    with open(os.path.join(tmp, "lib.cfg"), "w") as fo:
        fo.write(cfg.replace("@DIR@", tmp))
    with open(os.path.join(tmp, "pg.cfg"), "w") as fo:
        fo.write(pg_cfg)
    exe = os.path.join(REF, f"SOAPdenovo-Trans-{c['variant']}mer")
    out = os.path.join(tmp, "out")
    run = lambda args: subprocess.run([exe] + args, check=True, capture_output=True, text=True, timeout=120).stdout
    run(["pregraph", "-s", os.path.join(tmp, "pg.cfg"), "-K", str(c["K"]), "-p", str(c["p"]), "-o", out])
    run(["contig", "-g", out])
    log = run(["map", "-s", os.path.join(tmp, "lib.cfg"), "-g", out, "-p", str(c["p"])] + (["-r"] if c.get("trace") else [])
              + (["-f"] if c.get("fill") else []))
    info = dict(c)
    m = re.search(r"(\d+) nodes allocated, (\d+) kmer in reads, (\d+) kmer processed", log)
    info["nodes_allocated"], info["kmer_in_contigs"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"Output (\d+) out of (\d+)", log)
    info["reads_in_gap"], info["reads"] = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    m = re.search(r"(\d+) out of (\d+) \(", log.split("reads in gaps")[-1])
    info["reads_mapped"] = int(m.group(1))
    with open(os.path.join(cdir, "stdout.log"), "w") as fo:
        fo.write("".join(l + "\n" for l in log.splitlines() if "time spent" not in l and tmp not in l and "overall time" not in l))
    for f in files:
        with open(f, "rb") as fi, gzip.GzipFile(os.path.join(cdir, os.path.basename(f) + ".gz"), "wb", mtime=0) as fo:
            fo.write(fi.read())
    with open(os.path.join(cdir, "lib.cfg.template"), "w") as fo:
        fo.write(cfg)
    exts = ["contig", "ContigIndex", "preGraphBasic", "readOnContig", "ctg2Read", "peGrads", "readInGap"]
    if c.get("trace"):
        exts.append("readInformation")
    if c.get("fill"):                      # stored decompressed: the gzip bytes themselves are not part of the contract
        for ext in ("shortreadInGap", "PEreadOnContig"):
            with gzip.open(out + "." + ext + ".gz", "rb") as fi, open(out + "." + ext, "wb") as fo:
                fo.write(fi.read())
            exts.append(ext)
    for ext in exts:
        with open(out + "." + ext, "rb") as fi, gzip.GzipFile(os.path.join(cdir, "out." + ext + ".gz"), "wb", mtime=0) as fo:
            fo.write(fi.read())
    with open(os.path.join(cdir, "case.json"), "w") as fo:
        json.dump(info, fo, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print(c["name"], {k: info[k] for k in ("nodes_allocated", "kmer_in_contigs", "reads", "reads_mapped", "reads_in_gap")})


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if not only or c["name"] in only:
            make_case(c)
