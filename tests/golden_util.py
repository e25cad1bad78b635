"""helpers shared by the CPU and GPU parity tests: read the committed golden cases"""
from __future__ import annotations

import ctypes as C
import gzip
import json
import os

import numpy as np

import oracle_binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES_DIR = os.path.join(GOLD, "cases")
VARIANT_WORDS = {31: 1, 63: 2, 127: 4}
VARIANT_MAXK = {31: 31, 63: 63, 127: 127}


def case_names():
    return sorted(os.listdir(CASES_DIR))


def load_case(name):
    d = os.path.join(CASES_DIR, name)
    with open(os.path.join(d, "case.json")) as fi:
        info = json.load(fi)
    info["dir"] = d
    return info


def fastq_sequences(path):
    """sequence lines of a 4-line-record FASTQ (.gz)"""
    with gzip.open(path, "rb") as fi:
        lines = fi.read().split(b"\n")
    return lines[1::4][: len(lines) // 4]


def case_reads(info):
    """The read stream of a case in the reference's order (single file, or q1/q2 interleaved read1,read2,...
    prlHashReads.c:493-567), coded and truncated like readseqfq (readseq1by1.c:281-340).
    Returns (codes uint8[], offsets uint64[])."""
    L = ob.lib()
    d = info["dir"]
    if info["kind"] == "pe":
        a = fastq_sequences(os.path.join(d, "reads_1.fq.gz"))
        b = fastq_sequences(os.path.join(d, "reads_2.fq.gz"))
        seqs = [s for pair in zip(a, b) for s in pair]
    else:
        seqs = fastq_sequences(os.path.join(d, "reads.fq.gz"))
    max_rd_len = info["max_rd_len"]
    buf = np.zeros(max_rd_len + 8, dtype=np.uint8)
    out = []
    offs = [0]
    for s in seqs:
        n = L.sdto_encode_line(s, len(s), max_rd_len, buf.ctypes.data)
        out.append(buf[:n].copy())
        offs.append(offs[-1] + n)
    codes = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
    return codes, np.asarray(offs, dtype=np.uint64)


def golden_text(info, ext):
    with open(os.path.join(info["dir"], "out." + ext)) as fi:
        return fi.read()
