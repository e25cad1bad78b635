#!/bin/bash
# round 6, third GPU call: the whole -m gpu suite on the split tree + A/B of level-2 scatter variants on one batch per step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job3
mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --maxfail=6 > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
grep -h "skew_max_over_mean at" $O/pytest_gpu.txt
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in l2s2 l2wgs2 flush5; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
run "SDT_X=0" "--steps 3 --warmup 1 --track-first"
run "SDT_X=0" "--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000"
} 2>&1 | tee $O/ab.txt
