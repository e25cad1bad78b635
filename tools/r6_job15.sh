#!/bin/bash
# round 6, fifteenth GPU call: LDS tables of 1024 slots where the flush takes two slots per lane one after the other (2-word keys: 1280 slots,
# keys with ordinals: 1536): one memory round trip per flush instead of two, against 25 - 50 % more flushes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job15
mkdir -p $O
. tools/ab_env.sh
{
A="--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
run "SDT_X=0" "$A"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_nw2s1024.so" "$A"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_nw2s2048one.so" "$A"
run "SDT_X=0" "--steps 3 --warmup 1 --track-first"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_trk1024.so" "--steps 3 --warmup 1 --track-first"
} 2>&1 | tee $O/ab.txt
