"""soapdenovo-trans_amd -- MI355X-native `pregraph` hashing path of SOAPdenovo-Trans.

The product is ``csrc/libsdt_gpu.so`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/sdt_gpu.h``) plus the C host ``csrc/host`` that keeps the reference's
``pregraph -s cfg -K k -o out`` command line.  This Python package is the thin host-side mirror used by
tests and bench.py: it loads the shared library with ctypes and exposes one class whose methods have
the names and argument meaning of the C ABI.  There is no CPU fallback here: if the library is missing
or no gfx950 device is present the calls raise.

The directory name contains a hyphen, so import it through ``__graft_entry__.load_package()``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libsdt_gpu.so")
REPO_ROOT = os.path.dirname(PKG_DIR)

SDT_FLAG_DIRECT, SDT_FLAG_PARTITION, SDT_FLAG_TRACK_FIRST, SDT_FLAG_KEEP_READS, SDT_FLAG_CONTIG_INDEX = 1, 2, 4, 8, 16
SDT_OK, SDT_EINVAL, SDT_ENODEV, SDT_ENOMEM, SDT_EHIP, SDT_EFULL, SDT_ESTATE, SDT_ELIMIT = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/sdt_gpu.h declares: (name, restype, argtypes)
_c = ctypes
_ABI = [
    ("sdt_gpu_init", _c.c_int, [_c.POINTER(_c.c_void_p), _c.c_int, _c.c_int, _c.c_uint64, _c.c_uint32]),
    ("sdt_gpu_destroy", _c.c_int, [_c.c_void_p]),
    ("sdt_gpu_last_error", _c.c_char_p, []),
    ("sdt_gpu_abi_version", _c.c_int, []),
    ("sdt_gpu_reset", _c.c_int, [_c.c_void_p]),
    ("sdt_gpu_push_reads", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_push_reads_async", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_push_reads_fixed_async", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_uint64, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_push_wait", _c.c_int, [_c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_hint_total_kmers", _c.c_int, [_c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_host_alloc", _c.c_void_p, [_c.c_size_t]),
    ("sdt_gpu_host_free", None, [_c.c_void_p]),
    ("sdt_gpu_count_reads_device", _c.c_int,
     [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64, _c.c_uint64]),
    ("sdt_gpu_finish_count", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    ("sdt_shard_cut_ranges", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p]),
    ("sdt_shard_plan", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint32, _c.c_uint32] + [_c.c_void_p] * 6),
    ("sdt_kmer_final_bucket", _c.c_int, [_c.c_void_p, _c.c_int]),
    ("sdt_gpu_table_info", _c.c_int, [_c.c_void_p, _c.c_void_p]),
    ("sdt_sk_plan_count_items", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_uint64, _c.c_uint64, _c.c_uint32, _c.c_void_p, _c.c_uint32,
                                           _c.c_void_p, _c.c_void_p, _c.c_uint32, _c.c_void_p, _c.c_void_p]),
    ("sdt_gpu_delow", _c.c_int, [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_mark_and_hist", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_export_nodes", _c.c_int,
     [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64,
      _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_set_read_ordinal", _c.c_int, [_c.c_void_p, _c.c_uint64, _c.c_uint64]),
    ("sdt_gpu_load_paths", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_void_p,
                                      _c.c_uint64, _c.c_uint64]),
    ("sdt_gpu_export_paths", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_import_paths", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_uint64]),
    ("sdt_gpu_release_table", _c.c_int, [_c.c_void_p]),
    ("sdt_gpu_map_reads", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_export_arcs", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64,
                                       _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_set_node_index", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_update_nodes", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_tip_walks", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_tip_walks_compact", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_minor_out_dry", _c.c_int, [_c.c_void_p, _c.c_double, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64),
                                         _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_edge_ports", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_build_host_index", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_layout_sorted_keys", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_layout_apply", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_layout_on_device", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_export_ordered", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_update_nodes_by_index", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_tip_walks_labelled", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_minor_out_labelled", _c.c_int, [_c.c_void_p, _c.c_double, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_fetch_records", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_minor_out_commit", _c.c_int, [_c.c_void_p, _c.c_double, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("sdt_gpu_fetch_skipped", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_minor_out_commit_begin", _c.c_int, [_c.c_void_p, _c.c_double, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("sdt_gpu_minor_out_commit_finish", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("sdt_gpu_fetch_written", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_build_edges", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_fetch_edge_bases", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_index_contigs", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_set_contig_table", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_align_reads", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_int,
                                       _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_align_reads_device", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_uint64, _c.c_void_p,
                                              _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_key_words", _c.c_int, [_c.c_void_p]),
    ("sdt_gpu_table_slots", _c.c_uint64, [_c.c_void_p]),
    ("sdt_gpu_stream", _c.c_void_p, [_c.c_void_p]),
    ("sdt_gpu_set_stream", _c.c_int, [_c.c_void_p, _c.c_void_p]),
    ("sdt_gpu_kernel_time", _c.c_int,
     [_c.c_void_p, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    ("sdt_gpu_stage_times", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64)]),
    ("sdt_owner_hash", _c.c_uint64, [_c.c_void_p, _c.c_int]),
    ("sdt_gpu_comm_id", _c.c_int, [_c.c_void_p]),
    ("sdt_gpu_comm_init", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int]),
    ("sdt_gpu_comm_init_shm", _c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_int, _c.c_int]),
    ("sdt_gpu_count_reads_sharded", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64, _c.c_uint64]),
    ("sdt_gpu_push_reads_sharded", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_allreduce_i64", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int]),
    ("sdt_gpu_comm_stats", _c.c_int, [_c.c_void_p, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_double),
                                      _c.POINTER(_c.c_uint64)]),
    ("sdt_kmer_owner", _c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    ("sdt_kmer_bucket", _c.c_int, [_c.c_void_p, _c.c_int]),
    ("sdt_gpu_shard_ranges", _c.c_int, [_c.c_void_p, _c.c_void_p]),
    ("sdt_comm_selftest_shm", _c.c_int, [_c.c_char_p, _c.c_int, _c.c_int, _c.c_int]),
    ("sdt_gpu_import_nodes", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_uint64]),
    ("sdt_gpu_keep_reads", _c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_void_p, _c.c_uint64]),
]
ABI_SYMBOLS = [n for n, _, _ in _ABI]

_lib = None


class SdtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsdt_gpu error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU). Returns the library path."""
    cmd = ["make", "-C", CSRC_DIR] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load_library():
    """dlopen libsdt_gpu.so and bind every ABI symbol. Fails loudly when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("SDT_GPU_LIB", LIB_PATH)      # (A/B builds of the library: tools/, profiles/)
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the pregraph hashing path)")
    # One HIP runtime per process: torch bundles its own libamdhip64 (soname libamdhip64.so.7, file name
    # libamdhip64.so).  Loaded first, the dynamic linker resolves our NEEDED libamdhip64.so.7 to it; loaded
    # second, torch would map a second runtime next to /opt/rocm's and fail with hipErrorNoDevice.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in _ABI:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _ptr(a):
    """host numpy array or torch tensor (host or device) or int -> void*"""
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    return ctypes.c_void_p(a.data_ptr())  # torch tensor


def key_words_for(K: int) -> int:
    return 1 if K <= 31 else (2 if K <= 63 else 4)


def clamp_K(K: int, max_k: int = 127) -> int:
    """call_pregraph's K handling (pregraph.c:38-59): even -> +1, < 13 -> 13, > max -> max."""
    if K % 2 == 0:
        K += 1
    if K < 13:
        K = 13
    elif K > max_k:
        K = max_k
    return K


class PregraphGPU:
    """One device context: mirrors sdt_gpu_* one to one (see include/sdt_gpu.h for the reference
    call sites each method replaces)."""

    def __init__(self, K: int, est_distinct: int = 0, device: int = 0, flags: int = 0):
        self.lib = load_library()
        self.K = K
        self._ctx = ctypes.c_void_p()
        rc = self.lib.sdt_gpu_init(ctypes.byref(self._ctx), device, K, est_distinct, flags)
        self._check(rc)
        self.nw = self.lib.sdt_gpu_key_words(self._ctx)

    def _check(self, rc):
        if rc != SDT_OK:
            raise SdtError(rc, self.lib.sdt_gpu_last_error().decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.sdt_gpu_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- pass 1
    def reset(self):
        self._check(self.lib.sdt_gpu_reset(self._ctx))

    def push_reads(self, packed_words: np.ndarray, offsets: np.ndarray):
        packed_words = np.ascontiguousarray(packed_words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self.lib.sdt_gpu_push_reads(self._ctx, _ptr(packed_words), packed_words.size,
                                                _ptr(offsets), offsets.size - 1))

    def push_reads_async(self, packed_words: np.ndarray, offsets: np.ndarray) -> int:
        """enqueue only; the arrays must stay alive and untouched until push_wait(ticket) (pinned memory keeps the copy asynchronous)"""
        assert packed_words.dtype == np.uint32 and offsets.dtype == np.uint64 and packed_words.flags.c_contiguous and offsets.flags.c_contiguous
        t = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_push_reads_async(self._ctx, _ptr(packed_words), packed_words.size, _ptr(offsets), offsets.size - 1,
                                                      ctypes.byref(t)))
        return t.value

    def push_reads_fixed_async(self, packed_words: np.ndarray, nreads: int, read_len: int) -> int:
        """a batch of equal-length reads: the offsets are made on the device"""
        assert packed_words.dtype == np.uint32 and packed_words.flags.c_contiguous
        t = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_push_reads_fixed_async(self._ctx, _ptr(packed_words), packed_words.size, nreads, read_len, ctypes.byref(t)))
        return t.value

    def push_wait(self, ticket: int):
        self._check(self.lib.sdt_gpu_push_wait(self._ctx, ticket))

    def hint_total_kmers(self, kmers: int):
        self._check(self.lib.sdt_gpu_hint_total_kmers(self._ctx, kmers))

    def count_reads_device(self, d_words, nwords: int, d_offsets, nreads: int, max_read_len: int):
        self._check(self.lib.sdt_gpu_count_reads_device(self._ctx, _ptr(d_words), nwords, _ptr(d_offsets),
                                                        nreads, max_read_len))

    def finish_count(self):
        k, n = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_finish_count(self._ctx, ctypes.byref(k), ctypes.byref(n)))
        return k.value, n.value

    # -- multi-GPU, bucket sharding (include/sdt_gpu.h): every method below except kmer_owner is COLLECTIVE
    def comm_init(self, comm_id: bytes, rank: int, nranks: int):
        """RCCL communicator; comm_id = new_comm_id() of rank 0, handed to every rank"""
        buf = ctypes.create_string_buffer(bytes(comm_id), 128)
        self._check(self.lib.sdt_gpu_comm_init(self._ctx, buf, rank, nranks))

    def comm_init_shm(self, name: str, rank: int, nranks: int):
        """host shared-memory transport: validation where the ranks share one GPU"""
        self._check(self.lib.sdt_gpu_comm_init_shm(self._ctx, name.encode(), rank, nranks))

    def count_reads_sharded(self, d_words, nwords: int, d_offsets, nreads: int, max_read_len: int):
        self._check(self.lib.sdt_gpu_count_reads_sharded(self._ctx, _ptr(d_words) if nreads else None, nwords,
                                                         _ptr(d_offsets) if nreads else None, nreads, max_read_len))

    def push_reads_sharded(self, words: np.ndarray, offsets: np.ndarray):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self.lib.sdt_gpu_push_reads_sharded(self._ctx, _ptr(words), words.size, _ptr(offsets), offsets.size - 1))

    def allreduce(self, values) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.int64).copy()
        self._check(self.lib.sdt_gpu_allreduce_i64(self._ctx, _ptr(v), v.size))
        return v

    def shard_ranges(self, nranks: int) -> np.ndarray:
        """first level-1 bucket of every rank (+ the end): cut on the first sharded call"""
        a = np.zeros(nranks + 1, dtype=np.uint32)
        self._check(self.lib.sdt_gpu_shard_ranges(self._ctx, _ptr(a)))
        return a

    def comm_stats(self):
        """(bytes sent, bytes received, milliseconds on the exchange stream, exchanges) of this rank"""
        a, b, n, ms = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
        self._check(self.lib.sdt_gpu_comm_stats(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ms), ctypes.byref(n)))
        return a.value, b.value, ms.value, n.value

    # -- scans
    def delow(self, d: int) -> int:
        r = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_delow(self._ctx, d, ctypes.byref(r)))
        return r.value

    def mark_and_hist(self):
        hist = np.zeros(257, dtype=np.int64)
        lin = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_mark_and_hist(self._ctx, _ptr(hist), ctypes.byref(lin)))
        return hist, lin.value

    def export_nodes(self, with_first: bool = False):
        n = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_export_nodes(self._ctx, None, None, None, None, None, 0, ctypes.byref(n)))
        m = max(n.value, 1)
        keys = np.zeros((m, self.nw), dtype=np.uint64)
        l_links = np.zeros(m, dtype=np.uint32)
        r_flags = np.zeros(m, dtype=np.uint32)
        count = np.zeros(m, dtype=np.uint32)
        first = np.zeros(m, dtype=np.uint64) if with_first else None
        self._check(self.lib.sdt_gpu_export_nodes(self._ctx, _ptr(keys), _ptr(l_links), _ptr(r_flags),
                                                  _ptr(count), _ptr(first), m, ctypes.byref(n)))
        k = n.value
        if with_first:
            return keys[:k], l_links[:k], r_flags[:k], count[:k], first[:k]
        return keys[:k], l_links[:k], r_flags[:k], count[:k]

    # -- graph-cleaning dry runs on the device mirror (cutTipPreGraph.c)
    def set_node_index(self, keys: np.ndarray):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, self.nw)
        self._nidx = len(keys)
        self._check(self.lib.sdt_gpu_set_node_index(self._ctx, _ptr(keys), len(keys)))

    def update_nodes(self, keys, l_links, r_flags):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, self.nw)
        l_links = np.ascontiguousarray(l_links, dtype=np.uint32)
        r_flags = np.ascontiguousarray(r_flags, dtype=np.uint32)
        assert len(l_links) == len(keys) == len(r_flags)
        self._check(self.lib.sdt_gpu_update_nodes(self._ctx, _ptr(keys), _ptr(l_links), _ptr(r_flags), len(keys)))

    def tip_walks(self, thin: bool, cut_len: int):
        n = self._nidx
        end = np.zeros(max(n, 1), dtype=np.uint64)
        info = np.zeros(max(n, 1), dtype=np.uint8)
        self._check(self.lib.sdt_gpu_tip_walks(self._ctx, int(thin), cut_len, _ptr(end), _ptr(info), n))
        return end[:n], info[:n]

    def minor_out_dry(self, threshold: float):
        """-> (records uint64[n, 9], n_junctions): see sdt_gpu_minor_out_dry"""
        cap = max(self._nidx // 8, 1024)
        while True:
            rec = np.zeros((cap, 9), dtype=np.uint64)
            nj, nr = ctypes.c_uint64(), ctypes.c_uint64()
            rc = self.lib.sdt_gpu_minor_out_dry(self._ctx, ctypes.c_double(threshold), _ptr(rec), cap, ctypes.byref(nj), ctypes.byref(nr))
            if rc == SDT_EFULL and nr.value > cap:
                cap = nr.value
                continue
            self._check(rc)
            return rec[: nr.value], nj.value

    def edge_ports(self):
        """-> records uint64[n, 17]: see sdt_gpu_edge_ports"""
        cap = max(self._nidx // 4, 1024)
        while True:
            rec = np.zeros((cap, 17), dtype=np.uint64)
            nr = ctypes.c_uint64()
            rc = self.lib.sdt_gpu_edge_ports(self._ctx, _ptr(rec), cap, ctypes.byref(nr))
            if rc == SDT_EFULL and nr.value > cap:
                cap = nr.value
                continue
            self._check(rc)
            return rec[: nr.value]

    # -- map stage (prlContig2nodes / prlRead2Ctg); the context must be created with FLAG_CONTIG_INDEX
    def index_contigs(self, packed_words: np.ndarray, offsets: np.ndarray, ids: np.ndarray):
        packed_words = np.ascontiguousarray(packed_words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert len(ids) == len(offsets) - 1
        self._check(self.lib.sdt_gpu_index_contigs(self._ctx, _ptr(packed_words), packed_words.size, _ptr(offsets),
                                                   _ptr(ids), len(ids)))

    def set_contig_table(self, length: np.ndarray, twin: np.ndarray):
        length = np.ascontiguousarray(length, dtype=np.uint32)
        twin = np.ascontiguousarray(twin, dtype=np.uint32)
        assert len(length) == len(twin)
        self._check(self.lib.sdt_gpu_set_contig_table(self._ctx, _ptr(length), _ptr(twin), len(length) - 1))

    def align_reads(self, packed_words, offsets, align_len=None, align_len_all: int = 0, max_hits=None):
        """-> (read_info uint64[nreads], hits uint32[nhits, 4]); hits[r] = first hit of read r, further hits in the tail;
        see include/sdt_gpu.h for the bit layout"""
        packed_words = np.ascontiguousarray(packed_words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if align_len is not None:
            align_len = np.ascontiguousarray(align_len, dtype=np.int32)
            assert len(align_len) == n
        info = np.zeros(max(n, 1), dtype=np.uint64)
        cap = max_hits if max_hits is not None else 2 * n + 16
        while True:
            hits = np.zeros((max(cap, 1), 4), dtype=np.uint32)
            got = ctypes.c_uint64()
            rc = self.lib.sdt_gpu_align_reads(self._ctx, _ptr(packed_words), packed_words.size, _ptr(offsets), n,
                                              _ptr(align_len), align_len_all, _ptr(info), _ptr(hits), cap, ctypes.byref(got))
            if rc == SDT_EFULL and max_hits is None and got.value > cap:      # SDT_EFULL: the hit array was too small
                cap = got.value
                continue
            self._check(rc)
            return info[:n], hits[: got.value]

    def tip_walks_compact(self, thin: bool, cut_len: int):
        """-> records uint64[n, 2]: node index | info << 56, end index"""
        cap = max(self._nidx // 8, 1024)
        while True:
            rec = np.zeros((cap, 2), dtype=np.uint64)
            nr = ctypes.c_uint64()
            rc = self.lib.sdt_gpu_tip_walks_compact(self._ctx, int(thin), cut_len, _ptr(rec), cap, ctypes.byref(nr))
            if rc == SDT_EFULL and nr.value > cap:
                cap = nr.value
                continue
            self._check(rc)
            return rec[: nr.value]

    # -- the reference's visiting order on the device, labelled dry runs (include/sdt_gpu.h)
    def layout_sorted_keys(self, p: int, nw_variant: int):
        """-> (keys uint64[n, nw] sorted by (set, first occurrence), set_start uint64[p + 1])"""
        n = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_layout_sorted_keys(self._ctx, p, nw_variant, None, 0, None, ctypes.byref(n)))
        keys = np.zeros((max(n.value, 1), self.nw), dtype=np.uint64)
        ss = np.zeros(p + 1, dtype=np.uint64)
        self._check(self.lib.sdt_gpu_layout_sorted_keys(self._ctx, p, nw_variant, _ptr(keys), n.value, _ptr(ss), ctypes.byref(n)))
        return keys[: n.value], ss

    def layout_on_device(self, p: int, nw_variant: int, small_init: bool = False):
        """sort + replay of the probing + numbering, all on the device -> set_start uint64[p + 1]"""
        ss = np.zeros(p + 1, dtype=np.uint64)
        n = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_layout_on_device(self._ctx, p, nw_variant, int(small_init), _ptr(ss), ctypes.byref(n)))
        self._nidx = n.value
        return ss

    def layout_apply(self, order: np.ndarray):
        order = np.ascontiguousarray(order, dtype=np.uint64)
        self._nidx = len(order)
        self._check(self.lib.sdt_gpu_layout_apply(self._ctx, _ptr(order), len(order)))

    def export_ordered(self):
        n = self._nidx
        keys = np.zeros((max(n, 1), self.nw), dtype=np.uint64)
        l, r, cnt = (np.zeros(max(n, 1), dtype=np.uint32) for _ in range(3))
        self._check(self.lib.sdt_gpu_export_ordered(self._ctx, _ptr(keys), _ptr(l), _ptr(r), _ptr(cnt), n))
        return keys[:n], l[:n], r[:n], cnt[:n]

    def update_nodes_by_index(self, node, l_links, r_flags):
        node = np.ascontiguousarray(node, dtype=np.uint64)
        l_links = np.ascontiguousarray(l_links, dtype=np.uint32)
        r_flags = np.ascontiguousarray(r_flags, dtype=np.uint32)
        assert len(l_links) == len(node) == len(r_flags)
        self._check(self.lib.sdt_gpu_update_nodes_by_index(self._ctx, _ptr(node), _ptr(l_links), _ptr(r_flags), len(node)))

    def _fetch(self, n: int, words: int):
        rec = np.zeros((max(n, 1), words), dtype=np.uint64)
        self._check(self.lib.sdt_gpu_fetch_records(self._ctx, _ptr(rec), n * words))
        return rec[:n]

    def tip_walks_labelled(self, thin: bool, cut_len: int):
        """-> records uint64[n, 3]: node index | info << 56, end index, component label; sorted by (label, node)"""
        nr = ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_tip_walks_labelled(self._ctx, int(thin), cut_len, ctypes.byref(nr)))
        return self._fetch(nr.value, 3)

    def minor_out_labelled(self, threshold: float):
        """-> (records uint64[n, 14], n_junctions): node, 8 neighbours, 8 counts (two per word), label; the junction records sorted by (label, node)"""
        nj, nr = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_minor_out_labelled(self._ctx, ctypes.c_double(threshold), ctypes.byref(nj), ctypes.byref(nr)))
        return self._fetch(nr.value, 14), nj.value

    def minor_out_commit(self, threshold: float, max_component: int = 1 << 62):
        """removeMinorOut's commit on the device: labelled dry run + commit.  -> dict(records, n_junctions, largest, off, linear,
        node, l_links, r_flags, skipped, skipped_neighbours): the records of the dry run, the nodes the commit wrote, the junction
        records of the components it left to the caller and the records of the neighbours those may cut"""
        nj, nr = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_minor_out_labelled(self._ctx, ctypes.c_double(threshold), ctypes.byref(nj), ctypes.byref(nr)))
        big, off, lin, nw, nsk, nskr = (ctypes.c_uint64() for _ in range(6))
        self._check(self.lib.sdt_gpu_minor_out_commit(self._ctx, ctypes.c_double(threshold), max_component, ctypes.byref(big), ctypes.byref(off),
                                                      ctypes.byref(lin), ctypes.byref(nw), ctypes.byref(nsk), ctypes.byref(nskr)))
        node = np.zeros(max(nw.value, 1), dtype=np.uint64)
        l = np.zeros(max(nw.value, 1), dtype=np.uint32)
        r = np.zeros(max(nw.value, 1), dtype=np.uint32)
        self._check(self.lib.sdt_gpu_fetch_written(self._ctx, _ptr(node), _ptr(l), _ptr(r), nw.value))
        sk = np.zeros((max(nskr.value, 1), 14), dtype=np.uint64)
        self._check(self.lib.sdt_gpu_fetch_skipped(self._ctx, _ptr(sk), nskr.value))
        rec = self._fetch(nr.value, 14)
        return dict(records=rec, n_junctions=nj.value, largest=big.value, off=off.value, linear=lin.value,
                    node=node[:nw.value], l_links=l[:nw.value], r_flags=r[:nw.value], skipped=sk[:nsk.value], skipped_neighbours=sk[nsk.value:nskr.value])

    def build_edges(self):
        """-> (records uint64[n_edges, 4 + 2 nw], bases bytes, num_ed): see sdt_gpu_build_edges"""
        ne, ids, nb = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_build_edges(self._ctx, ctypes.byref(ne), ctypes.byref(ids), ctypes.byref(nb)))
        rec = self._fetch(ne.value, 4 + 2 * self.nw)
        buf = np.zeros(max(nb.value, 1), dtype=np.uint8)
        self._check(self.lib.sdt_gpu_fetch_edge_bases(self._ctx, _ptr(buf), nb.value))
        return rec, buf[: nb.value].tobytes(), ids.value

    def set_read_ordinal(self, base: int, stride: int = 1):
        self._check(self.lib.sdt_gpu_set_read_ordinal(self._ctx, base, stride))

    # -- introspection
    def table_slots(self) -> int:
        return self.lib.sdt_gpu_table_slots(self._ctx)

    def table_info(self) -> dict:
        """layout of the node table as it stands (sdt_gpu_table_info)"""
        a = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.sdt_gpu_table_info(self._ctx, a.ctypes.data))
        return {"layout": "flat", "slots": int(a[1]), "nodes": int(a[2])}

    def stream(self) -> int:
        return self.lib.sdt_gpu_stream(self._ctx) or 0

    def set_stream(self, hip_stream: int):
        self._check(self.lib.sdt_gpu_set_stream(self._ctx, ctypes.c_void_p(hip_stream)))

    def kernel_time(self, reset: bool = True):
        ms, launches, kmers = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.sdt_gpu_kernel_time(self._ctx, int(reset), ctypes.byref(ms), ctypes.byref(launches),
                                                 ctypes.byref(kmers)))
        return ms.value, launches.value, kmers.value

    def stage_times(self):
        """(ms per stage [direct, sk scatter, sk split, sk count, sk fold], pipeline counters) since the last kernel_time(reset=True)"""
        ms = (ctypes.c_double * 5)()
        cnt = (ctypes.c_uint64 * 20)()
        self._check(self.lib.sdt_gpu_stage_times(self._ctx, ms, cnt))
        names = ("merges", "lds_spills", "pool_direct", "early_flushes", "chunks_l1", "chunks_l2", "batches", "batch_kmers", "cnt_ticks_setup",
                 "cnt_ticks_fill", "cnt_ticks_count", "cnt_ticks_merge", "sc_ticks_stage", "sc_ticks_minima", "sc_ticks_starts",
                 "sc_ticks_emit", "distinct_records", "records", "distinct_record_kmers", "reserved1")
        return [float(x) for x in ms], dict(zip(names, (int(x) for x in cnt)))


def new_comm_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to every rank)"""
    buf = ctypes.create_string_buffer(128)
    if load_library().sdt_gpu_comm_id(buf) != SDT_OK:
        raise SdtError(SDT_EHIP, load_library().sdt_gpu_last_error().decode())
    return buf.raw


def count_plan(off2, kpre2, first_limit, limit, max_launches=4096):
    """the count stage's work items and launches for chunk lists off2 / kpre2 (csrc/sdt_count_plan.h): (items [n, 4] = c0, c1 | whole,
    first and last final bucket; first_item; launch_kmers)"""
    o = np.ascontiguousarray(off2, dtype=np.uint32)
    kp = np.ascontiguousarray(kpre2, dtype=np.uint64)
    nb = len(o) - 1
    assert len(kp) == nb + 1
    cap = nb + int(o[-1]) // 1024 + 2
    items = np.zeros((cap, 4), dtype=np.uint32)
    first = np.zeros(max_launches + 2, dtype=np.uint32)
    lk = np.zeros(max_launches + 2, dtype=np.uint64)
    ni, nl = ctypes.c_uint32(), ctypes.c_uint32()
    rc = load_library().sdt_sk_plan_count_items(o.ctypes.data, kp.ctypes.data, nb, first_limit, limit, max_launches, items.ctypes.data, cap,
                                                first.ctypes.data, lk.ctypes.data, len(first), ctypes.addressof(ni), ctypes.addressof(nl))
    if rc != 0:
        raise SdtError(rc, load_library().sdt_gpu_last_error().decode())
    return items[: ni.value].copy(), first[: nl.value + 1].copy(), lk[: nl.value].copy()


def shard_cut_ranges(mat: np.ndarray, nranks: int) -> np.ndarray:
    """bucket ranges of equal weight from the all-gathered chunk-list offsets (nranks x 257 uint32); host only"""
    m = np.ascontiguousarray(mat, dtype=np.uint32).reshape(nranks, 257)
    r = np.zeros(nranks + 1, dtype=np.uint32)
    rc = load_library().sdt_shard_cut_ranges(m.ctypes.data, nranks, r.ctypes.data)
    if rc != SDT_OK:
        raise SdtError(rc, load_library().sdt_gpu_last_error().decode())
    return r


def shard_plan(mat: np.ndarray, nranks: int, me: int, ranges: np.ndarray, recv_chunks: int, t: int):
    """sub-round t of the exchange as rank `me` sees it: (subrounds, send_begin, send_count, send_at, recv_count, recv_at)"""
    m = np.ascontiguousarray(mat, dtype=np.uint32).reshape(nranks, 257)
    rg = np.ascontiguousarray(ranges, dtype=np.uint32)
    S = ctypes.c_uint32()
    out = [np.zeros(nranks, dtype=np.uint32) for _ in range(5)]
    rc = load_library().sdt_shard_plan(m.ctypes.data, nranks, me, rg.ctypes.data, recv_chunks, t, ctypes.addressof(S),
                                       *[o.ctypes.data for o in out])
    if rc != SDT_OK:
        raise SdtError(rc, load_library().sdt_gpu_last_error().decode())
    return (S.value, *out)


def kmer_owner(key_words_msw_first, K: int, nranks: int) -> int:
    """rank that owns a canonical k-mer under bucket sharding (host copy of the device function)"""
    a = np.ascontiguousarray(key_words_msw_first, dtype=np.uint64)
    return load_library().sdt_kmer_owner(a.ctypes.data, K, nranks)


def kmer_bucket(key_words_msw_first, K: int) -> int:
    """level-1 minimizer bucket (0..255) of a canonical k-mer (host copy of the device function)"""
    a = np.ascontiguousarray(key_words_msw_first, dtype=np.uint64)
    return load_library().sdt_kmer_bucket(a.ctypes.data, K)


def kmer_final_bucket(key_words_msw_first, K: int) -> int:
    """final minimizer bucket (0 .. 2^18 - 1) of a canonical k-mer: the unit of the locality pipeline's count stage"""
    a = np.ascontiguousarray(key_words_msw_first, dtype=np.uint64)
    return load_library().sdt_kmer_final_bucket(a.ctypes.data, K)


def write_kmerfreq(path: str, hist) -> None:
    """freqStat (prlHashReads.c:994-1023): bins 1..255, one "%lld\\n" per line."""
    with open(path, "w") as fo:
        for i in range(1, 256):
            fo.write(f"{int(hist[i])}\n")


def kmerfreq_text(hist) -> str:
    return "".join(f"{int(hist[i])}\n" for i in range(1, 256))
