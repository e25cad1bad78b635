"""BASELINE.json configurations at FULL size on the GPU, through size-independent properties (no oracle can follow 6-9 G
k-mers in seconds): the two pass-1 kernel families -- one device atomic per occurrence, and minimizer buckets counted
in LDS -- must agree on every number the path reports (k-mers, nodes, `-d 1` removals, linear nodes, all 257 kmerFreq
bins: a checksum of checksums), the bins must add up to the nodes, the scan must be idempotent, and the oracle checks the
first 3 000 reads of the very same device buffers bit for bit.
  C2: 50 M x 150 bp, K = 31 (1-word keys)      C4: 50 M x 250 bp, K = 63 (2-word keys, the 127mer build's layout)"""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n,L,K,est", [("C2", 50_000_000, 150, 31, 300_000_000), ("C4", 50_000_000, 250, 63, 750_000_000)])
def test_baseline_config_at_full_size(pkg, synth, name, n, L, K, est):
    import torch
    dev = torch.device("cuda:0")
    words, offsets, nwords = synth.torch_workload(n, L, T=20000, device=dev, seed=42)
    torch.cuda.synchronize()
    seen = {}
    for mode in (pkg.SDT_FLAG_DIRECT, pkg.SDT_FLAG_PARTITION):
        with pkg.PregraphGPU(K, est_distinct=est, flags=mode) as g:
            g.count_reads_device(words, nwords, offsets, n, L)
            kmers, nodes = g.finish_count()
            assert kmers == n * (L - K + 1)
            hist0, linear0 = g.mark_and_hist()
            assert int(hist0.sum()) == nodes                       # every node lands in exactly one bin
            hist0b, linear0b = g.mark_and_hist()
            assert (hist0b == hist0).all() and linear0b == linear0  # the scan is idempotent
            removed = g.delow(1)                                   # -d 1, then the marks again (pregraph's order)
            hist1, linear1 = g.mark_and_hist()
            assert int(hist1.sum()) == nodes and 0 < removed < nodes
            seen[mode] = (kmers, nodes, removed, linear0, linear1, hist0.tolist(), hist1.tolist())
            if mode == pkg.SDT_FLAG_PARTITION:
                # the oracle on a slice of the same device buffers
                sub = 3000
                hw = words[: (sub * L + 15) // 16 + 1].cpu().numpy().view(np.uint32)
                idx = np.arange(sub * L)
                codes = ((hw[idx >> 4] >> (30 - 2 * (idx & 15)).astype(np.uint32)) & 3).astype(np.uint8)
                g.reset()
                g.count_reads_device(words, nwords, offsets, sub, L)
                k2, n2 = g.finish_count()
                o = ob.Oracle(K, nsets=8)
                o.add_reads(codes, (np.arange(sub + 1) * L).astype(np.uint64))
                assert (k2, n2) == (o.kmers_in_reads(), o.node_count())
                assert g.delow(1) == o.delow(1)
                oh, ol = o.mark()
                gh, gl = g.mark_and_hist()
                assert (gh == oh).all() and gl == ol
    assert seen[pkg.SDT_FLAG_DIRECT] == seen[pkg.SDT_FLAG_PARTITION], f"{name}: the two kernel families disagree"
