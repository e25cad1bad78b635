#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_run20
mkdir -p $O
bash tools/pmc_pipeline.sh $O/pmc200
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
grep -E "k_sk|k_mark|k_clear" $O/pmc200/kernel_stats.csv | awk -F'",' '{print substr($1,1,50), $2}'
tail -1 $O/pmc200/bench_under_rocprof.json | cut -c1-300
timeout 3000 python tools/e2e_pregraph.py --reads 20000000 --read-len 150 --K 31 --p 16 --T 20000 --ref-p2 8 --timeout 900 > $O/e2e_pregraph_20M_k31_p16_p8.json 2> $O/e2e_20M.err
tail -60 $O/e2e_pregraph_20M_k31_p16_p8.json
