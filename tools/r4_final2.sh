#!/bin/bash
# round 4, second final run -- after removeMinorOut's commit moved to the device (graph kernels are device sources: kernel_source_id
# changed): the whole -m gpu suite, kernel statistics + PMC passes + the judged bench line again, sdt-pregraph at 200 M reads (three
# runs) and at 20 M reads against the reference.  The matrix lines of r4_final.sh stand (pass 1 did not change).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_final2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
bash tools/pmc_pipeline.sh $O/pmc200 --est-distinct 809675638
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
cp $O/pmc200/kernel_stats.csv $O/kernel_stats_bench_200M_k31.csv 2>/dev/null; cp $O/pmc200/bench_under_rocprof.json $O/bench_under_rocprof_200M_k31.json 2>/dev/null
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
mkdir -p profiles/r4 && cp $O/pmc_pass1_200M_k31.json profiles/r4/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
timeout 900 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-300
timeout 500 python tools/e2e_pregraph.py --reads 200000000 --p 16 --T 20000 --skip-ref --timeout 150 --runs 3 > $O/e2e_pregraph_200M_k31_p16_ours_only.json 2> $O/e2e_200M.err
timeout 900 python tools/e2e_pregraph.py --reads 20000000 --p 16 --T 20000 --timeout 600 --runs 2 > $O/e2e_pregraph_20M_k31_p16.json 2> $O/e2e_20M.err
python3 - $O <<'E'
import json, sys
o = sys.argv[1]
for f in ("e2e_pregraph_200M_k31_p16_ours_only.json", "e2e_pregraph_20M_k31_p16.json"):
    try:
        d = json.load(open(o + "/" + f))
        print(f, d.get("ours_walls_s"), d.get("ref_wall_s"), d.get("identical"))
    except Exception as e:
        print(f, "FAILED", e)
E
