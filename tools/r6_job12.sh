#!/bin/bash
# round 6, twelfth GPU call: 32-byte slots for the level-2 records (a group of four = one aligned 128-byte line) now that the level-2
# scatter's rounds are cheaper -- round 5 measured this SLOWER when the kernel was bound by its phases
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job12
mkdir -p $O
SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rec2pad.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth" > $O/pytest_parity_rec2pad.txt 2>&1; echo "rec2pad parity: $(tail -1 $O/pytest_parity_rec2pad.txt)"
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rec2pad.so SDT_SK_POOL_MEM_PCT=88" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rec2pad.so" "--steps 3 --warmup 1"
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rec2pad.so" "--steps 3 --warmup 1 --reads 50000000"
} 2>&1 | tee $O/ab.txt
. tools/ab_env.sh
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_tuning.so SDT_SK_COUNT_ITEM_CHUNKS=2048" "--steps 3 --warmup 1" 2>&1 | tee -a gpurun_out/r6_job12/ab.txt
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_tuning.so SDT_SK_COUNT_ITEM_CHUNKS=512" "--steps 3 --warmup 1" 2>&1 | tee -a gpurun_out/r6_job12/ab.txt
