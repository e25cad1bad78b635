"""ctypes view of oracle/libsdt_oracle.so (the CPU restatement) -- test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libsdt_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


class Kmer(C.Structure):
    _fields_ = [("w", C.c_uint64 * 4)]

    @staticmethod
    def of(words4):
        k = Kmer()
        for i in range(4):
            k.w[i] = int(words4[i])
        return k

    def tup(self):
        return tuple(int(self.w[i]) for i in range(4))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("sdt_oracle.c", "sdt_oracle_graph.c", "sdt_oracle.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(LIB)
    L.sdto_base2int.restype = C.c_int
    L.sdto_base2int.argtypes = [C.c_int]
    L.sdto_encode_line.restype = C.c_int
    L.sdto_encode_line.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    for name in ("sdto_create_filter",):
        getattr(L, name).restype = Kmer
        getattr(L, name).argtypes = [C.c_int]
    L.sdto_next_kmer.restype = Kmer
    L.sdto_next_kmer.argtypes = [Kmer, C.c_int, C.c_int]
    L.sdto_prev_kmer.restype = Kmer
    L.sdto_prev_kmer.argtypes = [Kmer, C.c_int, C.c_int]
    L.sdto_reverse_complement.restype = Kmer
    L.sdto_reverse_complement.argtypes = [Kmer, C.c_int]
    L.sdto_kmer_smaller.restype = C.c_int
    L.sdto_kmer_smaller.argtypes = [Kmer, Kmer]
    L.sdto_first_char.restype = C.c_int
    L.sdto_first_char.argtypes = [Kmer, C.c_int]
    L.sdto_last_char.restype = C.c_int
    L.sdto_last_char.argtypes = [Kmer]
    L.sdto_hash_kmer.restype = C.c_uint64
    L.sdto_hash_kmer.argtypes = [Kmer, C.c_int]
    L.sdto_chop_read.restype = C.c_int
    L.sdto_chop_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sdto_next_prime.restype = C.c_uint64
    L.sdto_next_prime.argtypes = [C.c_uint64]
    L.sdto_set_new.restype = C.c_void_p
    L.sdto_set_new.argtypes = [C.c_uint64, C.c_float]
    L.sdto_set_free.argtypes = [C.c_void_p]
    L.sdto_set_put.restype = C.c_int
    L.sdto_set_put.argtypes = [C.c_void_p, Kmer, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.sdto_set_search.restype = C.c_int
    L.sdto_set_search.argtypes = [C.c_void_p, Kmer, C.c_int, C.POINTER(C.c_uint64)]
    L.sdto_set_first_probe.restype = C.c_uint64
    L.sdto_set_first_probe.argtypes = [C.c_void_p, Kmer, C.c_int]
    L.sdto_sets_new.restype = C.c_void_p
    L.sdto_sets_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.sdto_sets_new_a.restype = C.c_void_p
    L.sdto_sets_new_a.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.sdto_sets_free.argtypes = [C.c_void_p]
    L.sdto_sets_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.sdto_sets_node_count.restype = C.c_uint64
    L.sdto_sets_node_count.argtypes = [C.c_void_p]
    L.sdto_sets_delow.restype = C.c_uint64
    L.sdto_sets_delow.argtypes = [C.c_void_p, C.c_int]
    L.sdto_sets_mark.restype = C.c_uint64
    L.sdto_sets_mark.argtypes = [C.c_void_p, C.c_void_p]
    L.sdto_write_kmerfreq.restype = C.c_int
    L.sdto_write_kmerfreq.argtypes = [C.c_char_p, C.c_void_p]
    L.sdto_sets_export.restype = C.c_uint64
    L.sdto_sets_export.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.sdto_sets_export_first.restype = C.c_uint64
    L.sdto_sets_export_first.argtypes = [C.c_void_p, C.c_void_p]
    for name in ("sdto_remove_single_tips", "sdto_remove_minor_tips"):
        getattr(L, name).restype = C.c_uint64
        getattr(L, name).argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.sdto_remove_minor_out.restype = C.c_uint64
    L.sdto_remove_minor_out.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.sdto_write_vertex.restype = C.c_uint64
    L.sdto_write_vertex.argtypes = [C.c_void_p, C.c_char_p]
    L.sdto_node_set.restype = None
    L.sdto_node_set.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.sdto_tip_walk.restype = C.c_int
    L.sdto_tip_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    L.sdto_neighbours.restype = None
    L.sdto_neighbours.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sdto_minor_out_probe.restype = C.c_int
    L.sdto_minor_out_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    L.sdto_write_edges.restype = C.c_uint64
    L.sdto_write_edges.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.sdto_read2edge.restype = C.c_uint64
    L.sdto_read2edge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p]
    L.sdto_edge_port.restype = C.c_int
    L.sdto_edge_port.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib = L
    return L


class SetStruct(C.Structure):      # sdto_set
    _fields_ = [("array", C.c_void_p), ("flags", C.c_void_p), ("size", C.c_uint64), ("count", C.c_uint64),
                ("max", C.c_uint64), ("load_factor", C.c_double)]


class SetsStruct(C.Structure):     # sdto_sets
    _fields_ = [("nsets", C.c_int), ("nw", C.c_int), ("K", C.c_int), ("sets", C.c_void_p),
                ("kmers_in_reads", C.c_uint64), ("reads_seen", C.c_uint64)]


def key_words_for(K):
    return 1 if K <= 31 else (2 if K <= 63 else 4)


class Oracle:
    """prlRead2HashTable restated: nsets = thrd_num, nw = key words of the reference variant."""

    def __init__(self, K, nsets=8, nw=None, a=0):
        self.L = lib()
        self.K = K
        self.nw = nw or key_words_for(K)
        self.h = self.L.sdto_sets_new_a(nsets, self.nw, K, a)        # a = the reference's -a option

    def __del__(self):
        try:
            if self.h:
                self.L.sdto_sets_free(self.h)
                self.h = None
        except Exception:
            pass

    def add_reads(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.L.sdto_sets_add_reads(self.h, codes.ctypes.data, offsets.ctypes.data, offsets.size - 1)

    def kmers_in_reads(self):
        return C.cast(self.h, C.POINTER(SetsStruct)).contents.kmers_in_reads

    def node_count(self):
        return self.L.sdto_sets_node_count(self.h)

    def delow(self, d):
        return self.L.sdto_sets_delow(self.h, d)

    def mark(self):
        hist = np.zeros(257, dtype=np.int64)
        lin = self.L.sdto_sets_mark(self.h, hist.ctypes.data)
        return hist, lin

    # graph cleaning (cutTipPreGraph.c); each returns (count printed by the pass, "linear nodes" of its closing mark)
    def remove_minor_out(self, dd=5):
        ml = C.c_uint64()
        return self.L.sdto_remove_minor_out(self.h, dd, C.byref(ml)), ml.value

    def remove_single_tips(self):
        ml = C.c_uint64()
        return self.L.sdto_remove_single_tips(self.h, C.byref(ml)), ml.value

    def remove_minor_tips(self):
        ml = C.c_uint64()
        return self.L.sdto_remove_minor_tips(self.h, C.byref(ml)), ml.value

    def write_edges(self, path):
        """kmer2edges (mutates the graph as the reference does) -> (num_ed, edges emitted, extra (K+1)-mer nodes)"""
        ec, ex = C.c_uint64(), C.c_uint64()
        n = self.L.sdto_write_edges(self.h, path.encode(), C.byref(ec), C.byref(ex))
        return n, ec.value, ex.value

    def read2edge(self, codes, offs, path):
        """prlRead2edge after write_edges: text of *.preArc; returns the number of arcs"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        return self.L.sdto_read2edge(self.h, codes.ctypes.data, offs.ctypes.data, len(offs) - 1, path.encode())

    def write_vertex(self, path):
        return self.L.sdto_write_vertex(self.h, path.encode())

    # read-only probes of the cleaning passes (oracle/sdt_oracle_graph.c): the device dry runs are compared with them.
    # Keys are rows of 4 uint64, most significant first (the export layout).
    @staticmethod
    def _key4(key_row):
        a = np.zeros(4, dtype=np.uint64)
        kr = np.asarray(key_row, dtype=np.uint64).reshape(-1)
        a[4 - kr.size:] = kr
        return a

    def node_set(self, key_row, l_links, r_links, linear, deleted):
        k = self._key4(key_row)
        self.L.sdto_node_set(self.h, k.ctypes.data, int(l_links), int(r_links), int(linear), int(deleted))

    def tip_walk(self, key_row, cut_len, thin):
        """None, or (end key as a Python int, info = ch | sm << 2 | thin_stop << 3)"""
        k, e, info = self._key4(key_row), np.zeros(4, dtype=np.uint64), C.c_int()
        if not self.L.sdto_tip_walk(self.h, k.ctypes.data, cut_len, int(thin), e.ctypes.data, C.byref(info)):
            return None
        v = 0
        for x in e:
            v = (v << 64) | int(x)
        return v, info.value

    def neighbours(self, key_row):
        """8 entries (left links 0..3, right links 0..3): None, or (neighbour key as int, smaller)"""
        k, nb, st = self._key4(key_row), np.zeros((8, 4), dtype=np.uint64), np.zeros(8, dtype=np.int32)
        self.L.sdto_neighbours(self.h, k.ctypes.data, nb.ctypes.data, st.ctypes.data)
        out = []
        for i in range(8):
            if st[i] < 0:
                out.append(None)
            else:
                v = 0
                for x in nb[i]:
                    v = (v << 64) | int(x)
                out.append((v, int(st[i])))
        return out

    def edge_port(self, key_row, port):
        """kmer2edges' walk from one port: -1 (the node starts no edge), None (no link), or
        (far node key as int, arrival port, length, bal_edge)"""
        k, e, info = self._key4(key_row), np.zeros(4, dtype=np.uint64), np.zeros(3, dtype=np.int32)
        rc = self.L.sdto_edge_port(self.h, k.ctypes.data, int(port), e.ctypes.data, info.ctypes.data)
        if rc < 0:
            return -1
        if rc == 0:
            return None
        v = 0
        for x in e:
            v = (v << 64) | int(x)
        return v, int(info[0]), int(info[1]), int(info[2])

    def minor_out_probe(self, key_row, threshold):
        """cut flags of the 8 neighbours under clipKmerFromNode's ratio test on the graph as it is"""
        k, cut = self._key4(key_row), np.zeros(8, dtype=np.int32)
        n = self.L.sdto_minor_out_probe(self.h, k.ctypes.data, float(threshold), cut.ctypes.data)
        return n, [int(x) for x in cut]

    def export(self):
        n = self.node_count()
        keys = np.zeros((max(n, 1), 4), dtype=np.uint64)
        l = np.zeros(max(n, 1), dtype=np.uint32)
        r = np.zeros(max(n, 1), dtype=np.uint32)
        c = np.zeros(max(n, 1), dtype=np.uint32)
        f = np.zeros(max(n, 1), dtype=np.uint8)
        m = self.L.sdto_sets_export(self.h, keys.ctypes.data, l.ctypes.data, r.ctypes.data, c.ctypes.data, f.ctypes.data)
        assert m == n
        return keys[:n], l[:n], r[:n], c[:n], f[:n]

    def export_first(self):
        n = self.node_count()
        first = np.zeros(max(n, 1), dtype=np.uint64)
        assert self.L.sdto_sets_export_first(self.h, first.ctypes.data) == n
        return first[:n]


def kmerfreq_text(hist):
    return "".join(f"{int(hist[i])}\n" for i in range(1, 256))


def chop_read(codes, K, nw=None):
    """sdto_chop_read -> (keys uint64[n,4], prev uint8[n], next uint8[n], hash uint64[n])"""
    L = lib()
    nw = nw or key_words_for(K)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n = max(len(codes) - K + 1, 0)
    keys = np.zeros((max(n, 1), 4), dtype=np.uint64)
    p = np.zeros(max(n, 1), dtype=np.uint8)
    q = np.zeros(max(n, 1), dtype=np.uint8)
    h = np.zeros(max(n, 1), dtype=np.uint64)
    m = L.sdto_chop_read(codes.ctypes.data, len(codes), K, nw, keys.ctypes.data, p.ctypes.data, q.ctypes.data, h.ctypes.data)
    return keys[:m], p[:m], q[:m], h[:m]


def ref_binary(variant):
    """oracle/_ref/SOAPdenovo-Trans-<variant>mer if it was built (container with /root/reference)"""
    p = os.path.join(REF_DIR, f"SOAPdenovo-Trans-{variant}mer")
    return p if os.path.exists(p) else None
