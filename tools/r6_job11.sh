#!/bin/bash
# round 6, eleventh GPU call: 2 (default) / 3 / 4 records per lane and round in the level-2 scatter, blocks of 4096 ids
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job11
mkdir -p $O
for v in rpl4 rpl3; do
SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth" > $O/pytest_parity_$v.txt 2>&1; echo "$v parity: $(tail -1 $O/pytest_parity_$v.txt)"
done
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in rpl3 rpl4 rpl2blk4096; do
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_rpl4.so" "--steps 2 --warmup 1 --reads 400000000 --sigma 2.5 --d 1"
} 2>&1 | tee $O/ab.txt
