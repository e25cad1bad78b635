#!/bin/bash
# round 6, ninth GPU call: round 5's keeper (first new ticket) with blocks of 1024 ids against the keeper-by-index kernel, C3 and C5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job9
mkdir -p $O
. tools/ab_env.sh
{
A="--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_oldkeeper_blk1024.so" "$A"
run "SDT_X=0" "$A"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_oldkeeper_blk1024.so" "--steps 3 --warmup 1"
run "SDT_X=0" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_oldkeeper_blk1024.so" "--steps 3 --warmup 1 --sigma 2.5"
run "SDT_X=0" "--steps 3 --warmup 1 --sigma 2.5"
} 2>&1 | tee $O/ab.txt
