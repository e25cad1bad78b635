#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run46
mkdir -p $O
for u in 2 4 1; do
(cd soapdenovo-trans_amd/csrc && make -B libsdt_gpu.so EXTRA="-DSDT_SK_UNIT=$u" > /dev/null 2>&1)
timeout 900 python bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --extras 0 > $O/b_$u.log 2>$O/b.err
echo "unit $u: $(grep -o 'stage_ms_per_step[^}]*}' $O/b_$u.log) $(grep -o '"value": [0-9.]*' $O/b_$u.log | head -1) $(tail -2 $O/b.err | cut -c1-200)"
if [ $u != 1 ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t_$u.log 2>&1; grep -E "passed|failed" $O/t_$u.log | tail -1; fi
done
