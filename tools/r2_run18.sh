#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run18
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3
timeout 1500 python bench.py > $O/bench_default.log 2> $O/bench_default.err
tail -3 $O/bench_default.err | cut -c1-300; tail -1 $O/bench_default.log | cut -c1-1500
for n in 2 4; do
SDT_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2970$n bench.py --gpus $n --reads 40000000 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_share_$n.log 2> $O/bench_share_$n.err
tail -1 $O/bench_share_$n.log | cut -c1-300; grep -o '"exchange".*' $O/bench_share_$n.log | cut -c1-600
done
