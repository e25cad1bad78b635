#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run19
mkdir -p $O
df -h /dev/shm | tail -1
for n in 2 4; do
SDT_SHM_OUTBOX_MB=6000 SDT_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2970$n bench.py --gpus $n --reads 40000000 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_share_$n.log 2> $O/bench_share_$n.err
tail -1 $O/bench_share_$n.log | cut -c1-200; grep -o '"exchange".*' $O/bench_share_$n.log | cut -c1-700
grep "stage ms" $O/bench_share_$n.err | cut -c1-300
done
timeout 900 python bench.py --reads 40000000 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_40M.log 2> $O/bench_40M.err
tail -1 $O/bench_40M.log | cut -c1-200; grep -o '"pcie_inclusive".*' $O/bench_40M.log | cut -c1-300
