#!/bin/bash
# round 4: what depends on the device sources, taken again after their last change (kernel statistics, the three PMC passes, the
# judged bench line, a parity subset): same steps as in r4_final.sh, into gpurun_out/r4_final_pmc/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_final_pmc; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize.py -x -q -m gpu > $O/pytest_gpu_subset.log 2>&1; tail -1 $O/pytest_gpu_subset.log
bash tools/pmc_pipeline.sh $O/pmc200 --est-distinct 809675638
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
cp $O/pmc200/kernel_stats.csv $O/kernel_stats_bench_200M_k31.csv 2>/dev/null; cp $O/pmc200/bench_under_rocprof.json $O/bench_under_rocprof_200M_k31.json 2>/dev/null
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
mkdir -p profiles/r4 && cp $O/pmc_pass1_200M_k31.json profiles/r4/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
timeout 900 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-300
