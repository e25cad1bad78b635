"""Reproducer for the level-2 scatter geometry that loses records: the hot-bucket input through a
library built with -DSDT_SK_L2_LOG -DSDT_SK_L2_TPB=<lanes>, which logs every value a lane reads from a chunk cursor in
sk_reserve (slot granted / this lane opens the next chunk / look again / the exchange it wrote), every id sk_alloc_chunk hands out
and every slot k_sk_scatter_records stores a record into.  Per run: the k-mers counted; slots stored into twice; and, per
(workgroup, bucket) cursor, the first place where the sequence of values read from it breaks the protocol.
usage: SDT_GPU_LIB=<debug build> python tools/l2_lost_chunk.py [runs]"""
import collections, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
from soapdenovo_trans_amd.synth import pack_2bit
lib = pkg.load_library()
K, n, L = 21, 1500, 100
codes = np.zeros(n * L, dtype=np.uint8)
codes[L * 1000:] = np.tile(np.array([0, 1, 2, 3, 3, 1], dtype=np.uint8), (n - 1000) * L // 6 + 1)[: (n - 1000) * L]
offs = (np.arange(n + 1) * L).astype(np.uint64)
words = pack_2bit(codes)
CAP = 1 << 22
KIND = {0: "slot", 1: "OPENS-NEXT", 2: "again", 3: "WROTE"}
bad_runs = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    buf = torch.zeros(CAP, dtype=torch.int64, device="cuda:0")
    with pkg.PregraphGPU(K, est_distinct=1 << 16, flags=2) as g:
        lib.sdt_gpu_debug_l2_log.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        assert lib.sdt_gpu_debug_l2_log(g._ctx, buf.data_ptr(), CAP) == 0
        try:
            g.push_reads(words, offs)
            res = g.finish_count()
            err = ""
        except pkg.SdtError as e:
            res, err = None, str(e)
        lib.sdt_gpu_debug_l2_log(g._ctx, None, 0)
    h = buf.cpu().numpy().view(np.uint64)
    m = min(int(h[0]), CAP - 1)
    e = h[1: 1 + m]
    tag = (e >> np.uint64(61)).astype(np.int64)          # 0..3: a stored slot (bit 63 clear); 4 / 5: sk_alloc_chunk; 7: cursor event
    st = e[tag < 4]
    key = (st >> np.uint64(32)).astype(np.int64) * 64 + ((st >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
    uniq, cnt = np.unique(key, return_counts=True)
    dup_chunks = sorted(set((uniq[cnt > 1] // 64).tolist()))
    ok = res == (n * (L - K + 1), 7)
    bad_runs += not ok
    print(f"run {it}: {'ok ' if ok else 'BAD'} {res} {err[-70:]} | {len(st)} slots stored, {len(uniq[cnt > 1])} of them twice, in chunks {dup_chunks[:8]}; {m} log entries")
    if ok or not dup_chunks:
        continue
    cur = np.nonzero(tag == 7)[0]
    ce = e[cur]
    kind = ((ce >> np.uint64(59)) & np.uint64(3)).astype(np.int64)
    wg = ((ce >> np.uint64(55)) & np.uint64(15)).astype(np.int64)
    lane = ((ce >> np.uint64(45)) & np.uint64(1023)).astype(np.int64)
    lb = ((ce >> np.uint64(35)) & np.uint64(1023)).astype(np.int64)
    pos = ((ce >> np.uint64(30)) & np.uint64(31)).astype(np.int64)
    chunk = (ce & np.uint64(0x3FFFFFFF)).astype(np.int64)
    D = dup_chunks[0]
    hit = np.nonzero((chunk == D) & (kind != 2))[0]
    w0, b0 = int(wg[hit[0]]), int(lb[hit[0]])
    sel = np.nonzero((wg == w0) & (lb == b0))[0]
    print(f"   cursor (workgroup {w0}, bucket {b0}) of chunk {D}: {len(sel)} reads; in log order, without the look-agains:")
    seq = [(int(cur[i]), KIND[int(kind[i])], int(lane[i]), int(chunk[i]), int(pos[i])) for i in sel if kind[i] != 2]
    first = next(k for k, x in enumerate(seq) if x[3] == D)
    print("   " + " ".join(f"[{p}:{k}@{ln} {c}.{ps}]" for p, k, ln, c, ps in seq[max(first - 20, 0): first + 60]))
    again = collections.Counter(int(chunk[i]) for i in sel if kind[i] == 2)
    print("   look-agains per chunk value seen:", dict(sorted(again.items())[:12]))
print(bad_runs, "bad runs")
