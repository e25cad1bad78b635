#!/usr/bin/env python3
"""Throughput of the map stage's device kernels on a synthetic workload that is resident in HBM:
contigs = the transcriptome itself (T sequences), reads sampled from it (torch_workload).
    python tools/bench_align.py --reads 20000000 --read-len 150 --K 31 --steps 3
Prints one JSON line: reads/s, k-mers/s, ms per step of k_align_reads (HIP events inside the library), hits."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from soapdenovo_trans_amd import synth  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=20_000_000)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--K", type=int, default=31)
ap.add_argument("--T", type=int, default=20000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--err", type=float, default=0.002)
args = ap.parse_args()

dev = torch.device("cuda:0")
K, L, n = pkg.clamp_K(args.K), args.read_len, args.reads
codes, starts, _ = synth.make_transcriptome(args.T, seed=42)
ids = np.arange(1, 2 * args.T, 2, dtype=np.uint32)                 # contig ids 1, 3, 5, ... (each has a twin id + 1)
length = np.zeros(2 * args.T + 1, dtype=np.uint32)
twin = np.zeros(2 * args.T + 1, dtype=np.uint32)
lens = (starts[1:] - starts[:-1]).astype(np.uint32)
length[1::2], length[2::2] = lens, lens
twin[1::2], twin[2::2] = ids + 1, ids
t0 = time.time()
words, offsets, nwords = synth.torch_workload(n, L, args.T, dev, err=args.err)
torch.cuda.synchronize()
gen_s = time.time() - t0
with pkg.PregraphGPU(K, est_distinct=int(starts[-1]) + 1024, flags=pkg.SDT_FLAG_CONTIG_INDEX) as g:
    t0 = time.time()
    g.index_contigs(synth.pack_2bit(codes), starts.astype(np.uint64), ids)
    kmers_ctg, nodes = g.finish_count()
    index_s = time.time() - t0
    g.set_contig_table(length, twin)
    stream = torch.cuda.Stream()
    g.set_stream(stream.cuda_stream)
    info = torch.zeros(n, dtype=torch.int64, device=dev)
    cap = n + n // 4
    hits = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    got = ctypes.c_uint64()

    def step():
        rc = g.lib.sdt_gpu_align_reads_device(g._ctx, words.data_ptr(), offsets.data_ptr(), n, L, None, 32, info.data_ptr(),
                                              hits.data_ptr(), cap, ctypes.byref(got))
        if rc != 0:
            raise RuntimeError(g.lib.sdt_gpu_last_error().decode())

    for _ in range(args.warmup):
        step()
    g.kernel_time(reset=True)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = time.time() - t0
    kms, launches, _ = g.kernel_time(reset=True)
    nh = (info >> 40) & 255
    mapped = int((nh > 0).sum().item())
kmers = n * (L - K + 1)
print(json.dumps({"metric": "map stage: reads aligned to contigs / s", "reads": n, "read_len": L, "K": K, "contigs": args.T,
                  "contig_kmers": int(kmers_ctg), "contig_nodes": int(nodes), "index_s": round(index_s, 3), "steps": args.steps,
                  "ms_per_step": round(wall / args.steps * 1e3, 3), "kernel_ms_per_step": round(kms / max(args.steps, 1), 3),
                  "reads_per_s": round(n * args.steps / wall), "kmers_per_s": round(kmers * args.steps / wall),
                  "kernel_kmers_per_s": round(kmers * args.steps / (kms * 1e-3)) if kms else None,
                  "mapped_reads": mapped, "hits": int(got.value), "workload_gen_s": round(gen_s, 1)}))
