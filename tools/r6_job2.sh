#!/bin/bash
# round 6, second GPU call: the whole -m gpu suite on the tree without the node log + A/B of count-stage variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job2
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in flush5 flush6 l1bits9; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
run "SDT_SK_BATCH_LOG2=35 SDT_SK_POOL_MEM_PCT=80" "--steps 3 --warmup 1"
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
} 2>&1 | tee $O/ab.txt
