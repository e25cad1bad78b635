#!/usr/bin/env python3
"""What is there to gain from running the pipeline's stages side by side?  

Two library contexts, each with HALF of the judged workload and a node table of its own, driven (a) one after the other and
(b) by two host threads at once, each on its own stream -- the second started `--lag` ms after the first so that the level-1 /
level-2 scatters of one meet the count stage of the other, (c) the same on CU-masked streams
(hipExtStreamCreateWithCUMask: two disjoint halves of the chip).  The ratio (a) / (b) is what ANY interleaving of the
stages of consecutive batches inside one context could gain at most: the two contexts share nothing but the chip.

    python tools/overlap_probe.py --reads 200000000 [--lag 60] [--masks]
One JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=200_000_000)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--K", type=int, default=31)
ap.add_argument("--T", type=int, default=20000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lag", type=float, default=60.0, help="ms the second context starts after the first")
ap.add_argument("--masks", action="store_true", help="also run on two CU-masked streams")
ap.add_argument("--est", type=int, default=480_000_000, help="distinct k-mers expected per half")
args = ap.parse_args()

import torch  # noqa: E402

pkg = ge.load_package()
from soapdenovo_trans_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K, L = args.K, args.read_len
half = args.reads // 2 // 16 * 16
parts = []
for i in range(2):
    w, o, nw = synth.torch_workload(half, L, args.T, dev, seed=42 + 7 * i)
    parts.append((w, o, nw))
torch.cuda.synchronize()
kmers_half = half * (L - K + 1)

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(mask_words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(mask_words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: {rc}")
    return s.value


def step(ctx, part):
    w, o, nw = part
    ctx.reset()
    ctx.count_reads_device(w, nw, o, half, L)
    k, n = ctx.finish_count()
    ctx.mark_and_hist()
    assert k == kmers_half
    return n


def measure(streams, label, out):
    ctxs = [pkg.PregraphGPU(K, est_distinct=args.est, device=0) for _ in range(2)]
    try:
        for c, s in zip(ctxs, streams):
            c.set_stream(s)
        for c, p in zip(ctxs, parts):          # warm-up: pools, table
            step(c, p)
        torch.cuda.synchronize()
        seq, conc, solo = [], [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            step(ctxs[0], parts[0])
            t1 = time.perf_counter()
            step(ctxs[1], parts[1])
            t2 = time.perf_counter()
            seq.append(t2 - t0)
            solo.append([t1 - t0, t2 - t1])
            torch.cuda.synchronize()

            def run(i):
                if i:
                    time.sleep(args.lag * 1e-3)
                step(ctxs[i], parts[i])
            th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            conc.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        seq.sort(); conc.sort()
        out[label] = {"one_after_the_other_ms": [round(x * 1e3, 1) for x in seq], "side_by_side_ms": [round(x * 1e3, 1) for x in conc],
                      "solo_ms": [[round(a * 1e3, 1), round(b * 1e3, 1)] for a, b in solo],
                      "gain": round(seq[len(seq) // 2] / conc[len(conc) // 2], 3),
                      "rate_side_by_side_Gkmers_s": round(2 * kmers_half / conc[len(conc) // 2] / 1e9, 2),
                      "rate_one_after_the_other_Gkmers_s": round(2 * kmers_half / seq[len(seq) // 2] / 1e9, 2),
                      "stage_ms_ctx0": [round(x, 1) for x in ctxs[0].stage_times()[0]]}
    finally:
        for c in ctxs:
            c.close()


res = {"reads": 2 * half, "read_len": L, "K": K, "lag_ms": args.lag, "kmers": 2 * kmers_half}
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
measure([s0.cuda_stream, s1.cuda_stream], "two_plain_streams", res)
if args.masks:
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nwords = (ncu + 31) // 32
    lo = [0xFFFFFFFF if i < nwords // 2 else 0 for i in range(nwords)]
    hi = [0 if i < nwords // 2 else 0xFFFFFFFF for i in range(nwords)]
    ev = [0x55555555] * nwords
    od = [0xAAAAAAAA] * nwords
    for label, (a, b) in (("cu_mask_low_high", (lo, hi)), ("cu_mask_even_odd", (ev, od))):
        try:
            measure([masked_stream(a), masked_stream(b)], label, res)
        except Exception as e:  # noqa: BLE001
            res[label] = {"error": repr(e)}
print(json.dumps(res))
