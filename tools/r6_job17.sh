#!/bin/bash
# round 6, seventeenth GPU call: tick counters of the count stage (set-up / tile fill + scan / counting / merging) for 1-, 2- and 4-word keys
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job17
mkdir -p $O
for A in "--reads 200000000" "--reads 50000000 --read-len 250 --K 63" "--reads 50000000 --read-len 250 --K 95"; do
  SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_ticks.so timeout 600 python3 bench.py --steps 2 --warmup 1 --extras 0 --cpu-sample 0 $A 2>&1 | grep -E "stage ms per step" | cut -c1-900
done 2>&1 | tee $O/ticks.txt
