"""helpers for the `map` stage tests: read the committed cases under tests/golden/map_cases the way the reference's
`map` consumes them (test infrastructure; mirrors prlRead2Ctg.c:656-800 / readseq1by1.c:557-636,935-1131 for
well-formed paired inputs)"""
from __future__ import annotations

import ctypes as C
import gzip
import json
import os
import re

import numpy as np

import oracle_binding as ob

MAP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_cases")
VARIANT_WORDS = {31: 1, 63: 2, 127: 4}


def case_names():
    return sorted(os.listdir(MAP_DIR))


def load_case(name):
    d = os.path.join(MAP_DIR, name)
    with open(os.path.join(d, "case.json")) as fi:
        info = json.load(fi)
    info["dir"] = d
    return info


def gz_bytes(info, ext):
    with gzip.open(os.path.join(info["dir"], "out." + ext + ".gz"), "rb") as fi:
        return fi.read()


def materialise(info, tmp):
    """unpack the case into tmp: reads, lib.cfg and the graph files `map -g tmp/out` reads"""
    d = info["dir"]
    for f in os.listdir(d):
        if f.startswith("lib") and f.endswith(".gz"):
            with gzip.open(os.path.join(d, f), "rb") as fi, open(os.path.join(tmp, f[:-3]), "wb") as fo:
                fo.write(fi.read())
    for ext in ("contig", "ContigIndex", "preGraphBasic"):
        with open(os.path.join(tmp, "out." + ext), "wb") as fo:
            fo.write(gz_bytes(info, ext))
    cfg = os.path.join(tmp, "lib.cfg")
    with open(os.path.join(d, "lib.cfg.template")) as fi, open(cfg, "w") as fo:
        fo.write(fi.read().replace("@DIR@", str(tmp)))
    return cfg


def parse_cfg(text):
    """libraries in the reference's order (sorted by avg_ins, lib.c:437) with the fields `map` uses"""
    max_rd_len, libs, cur = 0, [], None
    for line in text.splitlines():
        line = line.strip()
        if line == "[LIB]":
            cur = dict(avg_ins=0, reverse=0, asm_flag=3, map_len=0, rank=0, pair_num_cut=0, rd_len_cutoff=0, files=[])
            libs.append(cur)
            continue
        if "=" not in line:
            continue
        k, v = line.split("=", 1)
        if k == "max_rd_len":
            max_rd_len = int(v)
        elif cur is None:
            continue
        elif k in ("f1", "f2", "q1", "q2", "p"):
            cur["files"].append((k, v))
        elif k == "reverse_seq":
            cur["reverse"] = int(v)
        elif k == "asm_flags":
            cur["asm_flag"] = int(v)
        elif k == "pair_num_cutoff":
            cur["pair_num_cut"] = int(v)
        elif k in cur:
            cur[k] = int(v)
    libs.sort(key=lambda l: l["avg_ins"])
    return max_rd_len or 100, libs


def _records(path, fastq):
    with gzip.open(path, "rb") as fi:
        lines = fi.read().split(b"\n")
    recs, cur = [], None
    if fastq:
        i = 0
        while i < len(lines):
            if lines[i].startswith(b"@"):
                seq = []
                i += 1
                while i < len(lines) and not lines[i].startswith(b"+"):
                    seq.append(lines[i])
                    i += 1
                i += 2                       # '+' line and the quality line
                recs.append(seq)
            else:
                i += 1
        return recs
    for l in lines:
        if l.startswith(b">"):
            cur = []
            recs.append(cur)
        elif cur is not None and not l.startswith(b"#") and l != b"":
            cur.append(l)
    return recs


def _encode(record_lines, max_len, reverse, buf):
    L = ob.lib()
    n, out = 0, []
    for l in record_lines:
        l = l + b"\n"
        m = L.sdto_encode_line(l, len(l), max_len - n, buf.ctypes.data)
        out.append(buf[:m].copy())
        n += m
    codes = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
    if reverse and len(codes):
        codes = (codes[::-1] ^ 2).astype(np.uint8)
    return codes


def case_reads(info):
    """(codes, offsets, lib_of_read, libs) in the order prlRead2Ctg consumes the reads: libraries by avg_ins, only
    asm_flags 2|3, only PAIRED inputs (f1/f2, q1/q2, p), mates alternating"""
    with open(os.path.join(info["dir"], "lib.cfg.template")) as fi:
        max_rd_len, libs = parse_cfg(fi.read())
    buf = np.zeros(max_rd_len + 8, dtype=np.uint8)
    out, offs, lib_of = [], [0], []
    for li, lib in enumerate(libs):
        if lib["asm_flag"] not in (2, 3):
            continue
        cut = min(lib["rd_len_cutoff"], max_rd_len) if lib["rd_len_cutoff"] > 0 else max_rd_len
        files = dict()
        for k, v in lib["files"]:
            files.setdefault(k, []).append(os.path.join(info["dir"], os.path.basename(v) + ".gz"))
        streams = []
        for a, b in zip(files.get("f1", []), files.get("f2", [])):          # curr_type 1
            ra, rb = _records(a, False), _records(b, False)
            streams.append([x for pair in zip(ra, rb) for x in pair])
        for a, b in zip(files.get("q1", []), files.get("q2", [])):          # curr_type 2
            ra, rb = _records(a, True), _records(b, True)
            streams.append([x for pair in zip(ra, rb) for x in pair])
        for a in files.get("p", []):                                        # curr_type 3
            streams.append(_records(a, False))
        for recs in streams:
            for r in recs:
                c = _encode(r, cut, lib["reverse"], buf)
                out.append(c)
                offs.append(offs[-1] + len(c))
                lib_of.append(li)
    codes = np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)
    return codes, np.asarray(offs, dtype=np.uint64), np.asarray(lib_of, dtype=np.int32), libs, max_rd_len


def case_contigs(info):
    """[(contig id, codes)] of out.contig in file order, after prlContig2nodes' length cut (K+2, prlHashCtg.c:343-350),
    plus K and the *.ContigIndex table"""
    basic = gz_bytes(info, "preGraphBasic").decode()
    K = int(re.search(r"VERTEX \d+ K (\d+)", basic).group(1))
    buf = np.zeros(1 << 16, dtype=np.uint8)
    ctgs = []
    recs = gz_bytes(info, "contig").split(b">")[1:]
    for i, rec in enumerate(recs, start=1):
        head, *seq = rec.split(b"\n")
        m = re.match(rb"(\d+)", head)
        cid = int(m.group(1)) if m else 0
        codes = _encode([s for s in seq if s], 1 << 30, 0, np.zeros(sum(len(s) for s in seq) + 8, dtype=np.uint8))
        if len(codes) < K + 1 or len(codes) < K + 2:
            continue
        ctgs.append((cid if cid > 0 else i, codes))
    lines = gz_bytes(info, "ContigIndex").decode().splitlines()
    num_all = int(lines[0].split()[1])
    rows = [tuple(int(x) for x in l.split()) for l in lines[2:] if l.strip()]
    return K, ctgs, num_all, np.array([r[1] for r in rows], dtype=np.uint32), np.array([r[2] for r in rows], dtype=np.int32)


class HitStruct(C.Structure):
    _fields_ = [("contigID", C.c_uint32), ("contigOffset", C.c_int32), ("readOffset", C.c_uint32),
                ("alignLength", C.c_uint32), ("orien", C.c_char)]


class MapOracle:
    """oracle/sdt_oracle_map.c"""

    def __init__(self, K, nsets, nw):
        self.L = L = ob.lib()
        L.sdto_map_new.restype = C.c_void_p
        L.sdto_map_new.argtypes = [C.c_int, C.c_int, C.c_int]
        L.sdto_map_free.argtypes = [C.c_void_p]
        L.sdto_map_set_contig_index.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.sdto_map_add_contig.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32]
        L.sdto_map_index_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.sdto_map_read.restype = C.c_int
        L.sdto_map_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sdto_map_run.restype = C.c_int
        L.sdto_map_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p]
        self.h = L.sdto_map_new(nsets, nw, K)
        self.K = K

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdto_map_free(self.h)
            self.h = None

    def set_contig_index(self, lengths, bals, num_all):
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        bals = np.ascontiguousarray(bals, dtype=np.int32)
        self.L.sdto_map_set_contig_index(self.h, lengths.ctypes.data, bals.ctypes.data, len(lengths), num_all)

    def add_contig(self, codes, cid):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.L.sdto_map_add_contig(self.h, codes.ctypes.data, len(codes), cid)

    def counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.sdto_map_index_counts(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def map_read(self, codes, align_len):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        hits = (HitStruct * 20)()
        best, foot = C.c_int(), C.c_int()
        n = self.L.sdto_map_read(self.h, codes.ctypes.data, len(codes), align_len, hits, C.byref(best), C.byref(foot))
        return n, [(h.contigID, h.contigOffset, h.readOffset, h.alignLength, h.orien.decode()) for h in hits[: max(n, 0)]], best.value, foot.value

    def run(self, codes, offs, lib_of, lib_ins, lib_map_len, max_read_len, p, prefix, buffer_size=100000000, trace=False, fill=False):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lib_of = np.ascontiguousarray(lib_of, dtype=np.int32)
        lib_ins = np.ascontiguousarray(lib_ins, dtype=np.int32)
        lib_map_len = np.ascontiguousarray(lib_map_len, dtype=np.int32)
        counters = np.zeros(4, dtype=np.int64)
        rc = self.L.sdto_map_run(self.h, codes.ctypes.data, offs.ctypes.data, len(offs) - 1, lib_of.ctypes.data,
                                 lib_ins.ctypes.data, lib_map_len.ctypes.data, max_read_len, p, buffer_size, int(trace), int(fill),
                                 str(prefix).encode(), counters.ctypes.data)
        assert rc == 0
        return [int(x) for x in counters]


def build_oracle(info):
    K, ctgs, num_all, lens, bals = case_contigs(info)
    o = MapOracle(K, info["p"], VARIANT_WORDS[info["variant"]])
    o.set_contig_index(lens, bals, num_all)
    for cid, codes in ctgs:
        o.add_contig(codes, cid)
    return o
