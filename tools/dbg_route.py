import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import synth
dev = torch.device('cuda:0')
K, L, n = 31, 150, 2_000_000
words, offsets, nwords = synth.torch_workload(n, L, 20000, dev, seed=42)
torch.cuda.synchronize()
g = pkg.PregraphGPU(K, est_distinct=1 << 26)
g.count_reads_device(words, nwords, offsets, n, L)
print('direct', g.finish_count())
g.reset()
total = n * (L - K + 1)
for nranks in (1, 2):
    cap = int(total / nranks * 1.25) + 4096
    recs = torch.zeros(cap * nranks * 2, dtype=torch.int64, device=dev)
    counts = torch.zeros(nranks, dtype=torch.int64, device=dev)
    displs = torch.zeros(nranks, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    g.extract_route(words, nwords, offsets, n, L, nranks, recs, cap * nranks, counts, displs)
    g.finish_count()
    torch.cuda.synchronize()
    c = counts.cpu().tolist(); d = displs.cpu().tolist()
    print('nranks', nranks, 'counts', c, 'sum', sum(c), 'expect', total, 'displs', d, 'cap', cap)
    r2 = recs.view(-1, 2)
    keys = torch.cat([r2[d[i]:d[i] + c[i], 0] for i in range(nranks)])
    print(' distinct keys in records', torch.unique(keys).numel(), 'zeros', int((keys == 0).sum()))
    g.reset()
    for i in range(nranks):
        sl = r2[d[i]:d[i] + c[i]].contiguous()
        g.insert_records(sl, c[i])
    print(' after insert', g.finish_count())
    g.reset()

# bench-like flow: rounds of 500k reads through one send buffer, world = 1, own stream
stream = torch.cuda.Stream(device=dev)
g.set_stream(stream.cuda_stream)
per = 500_000
cap = int(per * (L - K + 1) * 1.25) + 4096
send = torch.empty(cap * 2, dtype=torch.int64, device=dev)
counts = torch.zeros(1, dtype=torch.int64, device=dev)
displs = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
g.reset()
with torch.cuda.stream(stream):
    for r0 in range(0, n, per):
        g.extract_route(words, nwords, offsets[r0:], per, L, 1, send, cap, counts, displs)
        c = counts.cpu().tolist()
        g.insert_records(send, c[0])
print('rounds', g.finish_count())
