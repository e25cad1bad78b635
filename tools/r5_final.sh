#!/bin/bash
# round 5, the measurement runs (two gpurun calls: `r5_final.sh a` = what depends on the device sources -- the whole -m gpu suite, kernel
# statistics + PMC passes of the judged configuration with the flat merges (default) and with the node log, the judged bench line with
# SDT_TIMING on stderr; `r5_final.sh b` = the reporting matrix and the wall-clock runs of sdt-pregraph)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5_final; mkdir -p $O
if [ "$1" = "a" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
  bash tools/pmc_pipeline.sh $O/pmc200 --est-distinct 809675638
  python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
  cp $O/pmc200/kernel_stats.csv $O/kernel_stats_bench_200M_k31.csv 2>/dev/null; cp $O/pmc200/bench_under_rocprof.json $O/bench_under_rocprof_200M_k31.json 2>/dev/null
  find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
  mkdir -p profiles/r5 && cp $O/pmc_pass1_200M_k31.json profiles/r5/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
  SDT_PASS1_TABLE=log bash tools/pmc_pipeline.sh $O/pmc200log --est-distinct 809675638
  python3 tools/pmc_pipeline_summary.py $O/pmc200log 200000000 150 31 1 $O/pmc_nodelog_200M_k31.json
  cp $O/pmc200log/kernel_stats.csv $O/kernel_stats_bench_200M_k31_nodelog.csv 2>/dev/null
  find $O/pmc200log -name "pass_*" -type d | xargs rm -rf
  SDT_TIMING=1 timeout 900 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
  tail -1 $O/bench_default_200M_k31.json | cut -c1-300
  SDT_PASS1_TABLE=log timeout 600 python bench.py --cpu-sample 0 > $O/bench_nodelog_200M_k31.json 2> $O/bench_nodelog_200M_k31.err
  tail -1 $O/bench_nodelog_200M_k31.json | cut -c1-200
else
  run() { name=$1; shift; timeout 700 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json $name <<'E'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(j["value"] / 1e9, 2), "G k-mers/s", round(j["ms_per_step"], 1), "ms  frac", j["roofline"]["frac"], j["roofline"]["stage_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
E
  }
  run K23_100bp_200M --reads 200000000 --read-len 100 --K 23 --cpu-sample 4000000 --extras 0
  run K63_250bp_50M --reads 50000000 --read-len 250 --K 63 --cpu-sample 2000000 --extras 0
  run K95_250bp_50M --reads 50000000 --read-len 250 --K 95 --cpu-sample 0 --extras 0
  run C2_50M_k31 --reads 50000000 --cpu-sample 0 --extras 0
  run C5_400M_k31_d1_sigma2.5 --reads 400000000 --sigma 2.5 --d 1 --cpu-sample 0 --extras 0
  SDT_PASS1_TABLE=log run C2_50M_k31_nodelog --reads 50000000 --cpu-sample 0 --extras 0
  SDT_PASS1_TABLE=log run C5_400M_k31_d1_sigma2.5_nodelog --reads 400000000 --sigma 2.5 --d 1 --cpu-sample 0 --extras 0
  # wall clock: BASELINE's layout (paired-end files) at 200 M reads, ours only, three runs; 20 M reads against the reference on the same
  # box, and the same 20 M reads once more with --gpus 4 (four ranks on the one GPU: shared-memory transport)
  timeout 700 python tools/e2e_pregraph.py --reads 200000000 --p 16 --T 20000 --layout pe --skip-ref --timeout 200 --runs 4 --pause 20 > $O/e2e_pregraph_200M_k31_p16_pe_ours_only.json 2> $O/e2e_200M.err
  timeout 1200 python tools/e2e_pregraph.py --reads 20000000 --p 16 --T 20000 --layout pe --timeout 600 --runs 2 --also-cli-args "--gpus 4 --share-device" > $O/e2e_pregraph_20M_k31_p16_pe.json 2> $O/e2e_20M.err
  timeout 500 bash tools/e2e_profile.sh 200000000 pe $O/e2e_prof > $O/e2e_prof.log 2>&1; tail -12 $O/e2e_prof.log | cut -c1-130
  python3 - $O <<'E'
import json, sys
o = sys.argv[1]
for f in ("e2e_pregraph_200M_k31_p16_pe_ours_only.json", "e2e_pregraph_20M_k31_p16_pe.json"):
    try:
        d = json.load(open(o + "/" + f))
        print(f, d.get("ours_walls_s"), d.get("ref_wall_s"), d.get("identical"), d.get("also"))
    except Exception as e:
        print(f, "FAILED", e)
E
fi
