#!/usr/bin/env python3
"""PCIe-inclusive rate: the same pass 1 fed from HOST buffers through sdt_gpu_push_reads (H2D + kernel, double
buffered), as the C host does.  usage: bench_push.py [reads] [batch_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
K, L = 31, 150
dev = torch.device("cuda:0")
words, offsets, nwords = synth.torch_workload(n, L, 20000, dev, seed=42)
hw = words.cpu().numpy().view(np.uint32)
torch.cuda.synchronize()
wpb = batch * L // 16                                # batch*L is a multiple of 16
batches = []
for r0 in range(0, n, batch):
    nr = min(batch, n - r0)
    w0 = r0 * L // 16
    w = np.concatenate([hw[w0:w0 + (nr * L + 15) // 16], np.zeros(4, dtype=np.uint32)])
    batches.append((np.ascontiguousarray(w), (np.arange(nr + 1, dtype=np.uint64) * L)))
g = pkg.PregraphGPU(K, est_distinct=1 << 28)
for rep in range(2):
    g.reset(); g.finish_count()
    t0 = time.perf_counter()
    for w, o in batches:
        g.push_reads(w, o)
    kmers, nodes = g.finish_count()
    dt = time.perf_counter() - t0
    print(f"push_reads: {n} reads in batches of {batch}: {dt*1e3:.1f} ms -> {kmers/dt/1e9:.2f} G k-mers/s "
          f"({sum(w.nbytes + o.nbytes for w, o in batches)/dt/1e9:.1f} GB/s of host bytes), nodes {nodes}")
