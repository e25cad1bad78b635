import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import torch.distributed as dist
import __graft_entry__ as ge
pkg = ge.load_package()
from soapdenovo_trans_amd import sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for nrec in (10_000_000, 30_000_000, 60_000_000, 67_000_000, 67_200_000, 80_000_000, 134_300_000):
    cap = nrec + 100
    send = torch.arange(cap * 2, dtype=torch.int64, device=dev)
    recv = torch.full((cap * 2,), -1, dtype=torch.int64, device=dev)
    counts = torch.tensor([nrec], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    total, rc = sharding.exchange_records(send, counts, cap, 2, recv)
    torch.cuda.synchronize()
    dt = time.time() - t0
    ok = bool((recv[: nrec * 2] == send[: nrec * 2]).all())
    print(nrec, 'total', total, 'ok', ok, 'sec', round(dt, 3), 'GB/s', round(nrec * 16 / dt / 1e9, 1))
dist.destroy_process_group()
