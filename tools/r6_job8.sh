#!/bin/bash
# round 6, eighth GPU call: C5 (400 M reads, sigma = 2.5) lost 40 ms in the level-2 scatter with the keeper-by-index kernel: which change?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job8
mkdir -p $O
. tools/ab_env.sh
{
A="--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
run "SDT_X=0" "$A"
for v in blk128 oldl2; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "$A"
done
run "SDT_X=0" "$A"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_oldl2.so" "--steps 3 --warmup 1"
} 2>&1 | tee $O/ab.txt
