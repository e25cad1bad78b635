#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

  * unit_<31|63|127>.txt : output of oracle/_ref/probe<variant> (our driver linked against the
    reference's own kmer.o / hashFunction.o / newhash.o) -- unit vectors for the L3 primitives.
  * cases/<name>/        : seeded synthetic FASTQ (gzip), the config template, and what
    oracle/_ref/SOAPdenovo-Trans-<variant>mer pregraph wrote for it: kmerFreq, preGraphBasic, vertex,
    edge (gunzipped), preArc, plus the counters it printed.

Run from the repo root after `make -C oracle ref` (needs /root/reference; the GPU box never runs this).
Fixtures are data: inputs + the reference's outputs.  This script is the committed recipe.
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


def run_probe(variant):
    out = subprocess.run([os.path.join(REF, f"probe{variant}"), "24", "6000"], check=True, capture_output=True, text=True).stdout
    with open(os.path.join(HERE, f"unit_{variant}.txt"), "w") as fo:
        fo.write(out)
    print(f"unit_{variant}.txt: {len(out.splitlines())} lines")


def dirty(letters: bytes, rng) -> bytes:
    """sprinkle the characters the reference's parser treats specially (readseq1by1.c:296-326)"""
    b = bytearray(letters)
    for i in range(len(b)):
        u = rng.random()
        if u < 0.01:
            b[i] = ord("N")
        elif u < 0.015:
            b[i] = ord(".")
        elif u < 0.05:
            b[i] = b[i] + 32      # lowercase
        elif u < 0.052:
            b[i] = ord("n")
        elif u < 0.054:
            b[i] = ord("R")       # IUPAC letter: coded by its bits
    return bytes(b)


def write_fastq_raw(path, seqs):
    blob = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    if len(blob) % 32768 == 0:
        blob = b"@x" + blob[1:]
    with open(path, "wb") as fo:
        fo.write(blob)


CASES = [
    # name, variant, K, p, d, reads spec
    dict(name="se100_k23_p8", variant=31, K=23, p=8, d=0, n=4000, L=100, T=30, kind="se"),
    dict(name="se100_k23_p1", variant=31, K=23, p=1, d=0, n=4000, L=100, T=30, kind="se"),
    dict(name="se100_k23_p8_d1", variant=31, K=23, p=8, d=1, n=4000, L=100, T=30, kind="se"),
    dict(name="pe150_k31_p8", variant=31, K=31, p=8, d=0, n=1500, L=150, T=30, kind="pe"),
    dict(name="se250_k63_p8_127mer", variant=127, K=63, p=8, d=0, n=1500, L=250, T=30, kind="se"),
    dict(name="se150_k47_p4_63mer", variant=63, K=47, p=4, d=0, n=2000, L=150, T=30, kind="se"),
    dict(name="se150_k95_p3_127mer_d2", variant=127, K=95, p=3, d=2, n=2000, L=150, T=30, kind="se"),
    # -a <n != 0>: the 63mer / 127mer binaries start every set at init_kmerset(k * 0xFFFFFF) with k == 0, i.e. 3 slots
    # (prlHashReads.c:404-413): another growth sequence, another visiting order
    dict(name="se150_k47_p4_63mer_a1", variant=63, K=47, p=4, d=0, n=2000, L=150, T=30, kind="se", a=1),
    dict(name="dirty_ragged_k25_cut80", variant=31, K=25, p=8, d=0, n=3000, L=120, T=30, kind="dirty", max_rd_len=80),
    dict(name="evenK24_p2", variant=31, K=24, p=2, d=0, n=1500, L=100, T=20, kind="se"),
    dict(name="smallK11_p8", variant=31, K=11, p=8, d=0, n=800, L=60, T=10, kind="se"),
    # K = 127: the (K+1)-mers of length-1 edges are 128 bases long, where the reference's reverseComplement misfires
    # (its length parameter is a char); *.preArc shows it
    dict(name="alleles250_k127_p5_127mer", variant=127, K=127, p=5, d=0, n=6000, L=250, T=5, kind="alleles"),
]


def make_case(c):
    cdir = os.path.join(HERE, "cases", c["name"])
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    tx = synth.make_transcriptome(c["T"], seed=100 + len(c["name"]))
    tmp = tempfile.mkdtemp(prefix="sdtgold_")
    files = []
    max_rd_len = c.get("max_rd_len", c["L"])
    if c["kind"] == "se":
        codes, offs = synth.sample_reads(*tx, n_reads=c["n"], read_len=c["L"], seed=5, err=0.004)
        p = os.path.join(tmp, "reads.fq")
        synth.write_fastq(p, codes, offs)
        files = [p]
        cfg = f"max_rd_len={max_rd_len}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nq=@DIR@/reads.fq\n"
    elif c["kind"] == "pe":
        (c1, o1), (c2, o2) = synth.sample_pairs(*tx, n_pairs=c["n"], read_len=c["L"], seed=5, err=0.004)
        p1, p2 = os.path.join(tmp, "reads_1.fq"), os.path.join(tmp, "reads_2.fq")
        synth.write_fastq(p1, c1, o1)
        synth.write_fastq(p2, c2, o2)
        files = [p1, p2]
        cfg = f"max_rd_len={max_rd_len}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nq1=@DIR@/reads_1.fq\nq2=@DIR@/reads_2.fq\n"
    elif c["kind"] == "alleles":
        # three alleles of every transcript: the original, one with substitutions at positions p, one with substitutions
        # at p + 1 -- equally covered, so both branchings survive the cleaning and sit on ADJACENT k-mers: length-1 edges
        codes0, starts0, _ = tx
        seqs = []
        for t in range(len(starts0) - 1):
            a = codes0[starts0[t]:starts0[t + 1]].copy()
            b, d = a.copy(), a.copy()
            for pos in range(300, len(a) - 300, 350):
                b[pos] = (b[pos] + 1) & 3
                d[pos + 1] = (d[pos + 1] + 2) & 3
            seqs += [a, b, d]
        st = np.zeros(len(seqs) + 1, dtype=np.int64)
        st[1:] = np.cumsum([len(x) for x in seqs])
        w3 = np.array([float(len(x)) for x in seqs])
        tx3 = (np.concatenate(seqs), st, w3 / w3.sum())
        codes, offs = synth.sample_reads(*tx3, n_reads=c["n"], read_len=c["L"], seed=5, err=0.001)
        p = os.path.join(tmp, "reads.fq")
        synth.write_fastq(p, codes, offs)
        files = [p]
        cfg = f"max_rd_len={max_rd_len}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nq=@DIR@/reads.fq\n"
    else:  # dirty ragged single-end
        rng = np.random.default_rng(77)
        codes, offs = synth.sample_reads(*tx, n_reads=c["n"], read_len=c["L"], seed=5, err=0.004, ragged=True)
        letters = synth.BASES[codes].tobytes()
        o = offs.astype(np.int64)
        seqs = [dirty(letters[o[i]:o[i + 1]], rng) for i in range(len(o) - 1)]
        p = os.path.join(tmp, "reads.fq")
        write_fastq_raw(p, seqs)
        files = [p]
        cfg = f"max_rd_len={max_rd_len}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nq=@DIR@/reads.fq\n"
    with open(os.path.join(tmp, "lib.cfg"), "w") as fo:
        fo.write(cfg.replace("@DIR@", tmp))
    exe = os.path.join(REF, f"SOAPdenovo-Trans-{c['variant']}mer")
    cmd = [exe, "pregraph", "-s", os.path.join(tmp, "lib.cfg"), "-K", str(c["K"]), "-p", str(c["p"]), "-o", os.path.join(tmp, "out")]
    if c["d"]:
        cmd += ["-d", str(c["d"])]
    if c.get("a"):
        cmd += ["-a", str(c["a"])]
    log = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600).stdout
    m = re.search(r"(\d+) nodes allocated, (\d+) kmer in reads, (\d+) kmer processed", log)
    info = dict(c)
    info["nodes_allocated"], info["kmer_in_reads"], info["kmer_processed"] = (int(x) for x in m.groups())
    m = re.search(r"(\d+) linear nodes", log)
    info["linear_nodes"] = int(m.group(1))
    m = re.search(r"(\d+) kmer removed", log)
    info["kmer_removed"] = int(m.group(1)) if m else None
    info["max_rd_len"] = max_rd_len
    # graph-cleaning counters (cutTipPreGraph.c:367,434,1074; Mark1in1outNode :1227 prints after each pass)
    info["kmers_off"] = int(re.search(r"(\d+) kmers off", log).group(1))
    info["tips_off"] = [int(x) for x in re.findall(r"(\d+) tips off", log)]
    info["linear_after"] = [int(x) for x in re.findall(r"(\d+) linear nodes", log)]
    info["vertex_outputed"] = int(re.search(r"(\d+) vertex outputed", log).group(1))
    with open(os.path.join(cdir, "stdout.log"), "w") as fo:
        fo.write("".join(l + "\n" for l in log.splitlines() if "time spent" not in l and tmp not in l and "overall time" not in l))
    # fixtures
    for f in files:
        with open(f, "rb") as fi, gzip.GzipFile(os.path.join(cdir, os.path.basename(f) + ".gz"), "wb", mtime=0) as fo:
            fo.write(fi.read())
    with open(os.path.join(cdir, "lib.cfg.template"), "w") as fo:
        fo.write(cfg)
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        shutil.copy(os.path.join(tmp, "out." + ext), os.path.join(cdir, "out." + ext))
    with gzip.open(os.path.join(tmp, "out.edge.gz"), "rb") as fi, gzip.GzipFile(os.path.join(cdir, "out.edge.txt.gz"), "wb", mtime=0) as fo:
        fo.write(fi.read())
    with open(os.path.join(cdir, "case.json"), "w") as fo:
        json.dump(info, fo, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print(c["name"], {k: info[k] for k in ("nodes_allocated", "kmer_in_reads", "linear_nodes", "kmer_removed")})


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only:
        for v in (31, 63, 127):
            run_probe(v)
    for c in CASES:
        if not only or c["name"] in only:
            make_case(c)
