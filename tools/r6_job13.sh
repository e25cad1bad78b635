#!/bin/bash
# round 6, thirteenth GPU call: level-2 chunks of 32 records (half the chunks: chunk lists, allocations, k_sk_chunk_place)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job13
mkdir -p $O
SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_cap2_32.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth or first" > $O/pytest_parity_cap2_32.txt 2>&1; echo "cap2_32 parity: $(tail -1 $O/pytest_parity_cap2_32.txt)"
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_cap2_32.so" "--steps 3 --warmup 1"
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_cap2_32.so" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_cap2_32.so" "--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
} 2>&1 | tee $O/ab.txt
