#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run45
mkdir -p $O
timeout 900 python bench.py --reads 50000000 --steps 2 --warmup 1 --cpu-sample 0 --extras 0 > $O/b.log 2>$O/b.err
echo "$(grep -o 'stage_ms_per_step[^}]*}' $O/b.log) $(grep -o '"value": [0-9.]*' $O/b.log | head -1)"
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t.log 2>&1
grep -E "passed|failed" $O/t.log | tail -2
