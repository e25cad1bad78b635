#!/bin/bash
# round 6, seventh GPU call: blocks of chunk ids of the level-2 scatter (128 / 1024 / 4096), parity of the default first
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job7
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_sharded.py -q -x > $O/pytest_parity.txt 2>&1; tail -2 $O/pytest_parity.txt
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in blk128 blk4096; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
run "SDT_X=0" "--steps 3 --warmup 1"
} 2>&1 | tee $O/ab.txt
