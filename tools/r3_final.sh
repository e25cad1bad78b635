#!/bin/bash
# round 3, final measurements on the GPU box: the whole -m gpu suite, the kernel statistics + PMC passes (HBM traffic, SQ / LDS
# counters) of the judged configuration with the shipped build, and the judged bench line.  Everything under its own timeout.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3_final; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
# (--est-distinct = what bench.py's own estimate gives for this workload: its four prefix passes would otherwise be profiled too)
bash tools/pmc_pipeline.sh $O/pmc200 --est-distinct 809675638
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
cp $O/pmc200/kernel_stats.csv $O/kernel_stats_bench_200M_k31.csv 2>/dev/null; cp $O/pmc200/bench_under_rocprof.json $O/ 2>/dev/null
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
bash tools/pmc_sq.sh $O/sq --est-distinct 809675638
python3 tools/pmc_sq_summary.py $O/sq 24000000000 $O/sq_pass1_200M_k31.json > $O/sq_summary.txt 2>&1
find $O/sq -name "pass_*" -type d | xargs rm -rf
mkdir -p profiles/r3 && cp $O/pmc_pass1_200M_k31.json profiles/r3/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
timeout 900 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-300
