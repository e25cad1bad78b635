#!/bin/bash
# round 6, sixth GPU call: the level-2 scatter with one keeper lane per sub-bucket and D records per lane and round: parity first, then the rate
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job6
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_spills.py -q -x > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
for v in rpl4 rpl1; do
  SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "node_table_equals_oracle or golden_case or hot_bucket or growth" > $O/pytest_parity_$v.txt 2>&1; echo "$v parity: $(tail -1 $O/pytest_parity_$v.txt)"
done
. tools/ab_env.sh
{
run "SDT_X=0" "--steps 3 --warmup 1"
for v in rpl4 rpl1; do
  run "SDT_GPU_LIB=$PWD/gpurun_ab/libsdt_gpu_$v.so" "--steps 3 --warmup 1"
done
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 63"
run "SDT_X=0" "--steps 3 --warmup 1 --reads 50000000 --read-len 250 --K 95"
} 2>&1 | tee $O/ab.txt
