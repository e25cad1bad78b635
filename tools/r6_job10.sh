#!/bin/bash
# round 6, tenth GPU call: runs of chunk ids out of the workgroup's block (no global atomic per run): parity, then C3 at sigma 2 / 2.5 and C5
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job10
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_spills.py -q -x > $O/pytest_parity.txt 2>&1; grep -E "passed|failed" $O/pytest_parity.txt | tail -1
SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rpl2.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth" > $O/pytest_parity_rpl2.txt 2>&1; echo "rpl2 parity: $(tail -1 $O/pytest_parity_rpl2.txt)"
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rpl2.so" "--steps 3 --warmup 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_oldkeeper_blk1024.so" "--steps 3 --warmup 1"
run "SDT_X=0" "--steps 3 --warmup 1 --sigma 2.5"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rpl2.so" "--steps 3 --warmup 1 --sigma 2.5"
run "SDT_X=0" "--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rpl2.so" "--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
} 2>&1 | tee $O/ab.txt
