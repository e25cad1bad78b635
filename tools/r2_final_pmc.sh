#!/bin/bash
# re-take the judged bench line and the PMC passes after a change of the device sources (the rest of tools/r2_final.sh stays valid)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_final2
rm -rf $O; mkdir -p $O
bash tools/pmc_pipeline.sh $O/pmc200
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
mkdir -p profiles/r2 && cp $O/pmc_pass1_200M_k31.json profiles/r2/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
timeout 1800 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-400
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
