#!/bin/bash
# round 6, the measurement run (one gpurun call): the whole -m gpu suite, kernel statistics + PMC passes of the judged configuration, the judged
# bench line (with the e2e leg), the reporting matrix, sdt-pregraph at 200 M paired-end reads and at 20 M reads against the reference
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_final; mkdir -p $O profiles/r6
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/pmc_pipeline.sh $O/pmc200 --est-distinct 809675638
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
cp $O/pmc200/kernel_stats.csv $O/kernel_stats_bench_200M_k31.csv 2>/dev/null; cp $O/pmc200/bench_under_rocprof.json $O/bench_under_rocprof_200M_k31.json 2>/dev/null
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
cp $O/pmc_pass1_200M_k31.json profiles/r6/pmc_pass1_200M_k31.json      # bench.py reads the traffic from here
SDT_TIMING=1 timeout 900 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-400
run() { name=$1; shift; timeout 700 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json $name <<'E'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(j["value"] / 1e9, 2), "G k-mers/s", round(j["ms_per_step"], 1), "ms  frac", j["roofline"]["frac"], j["roofline"].get("as_127mer_build", {}).get("frac"), j["roofline"]["stage_ms_per_step"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
E
}
run K23_100bp_200M --reads 200000000 --read-len 100 --K 23 --cpu-sample 0 --extras 0
run K63_250bp_50M --reads 50000000 --read-len 250 --K 63 --cpu-sample 2000000 --extras 0
run K95_250bp_50M --reads 50000000 --read-len 250 --K 95 --cpu-sample 0 --extras 0
run C2_50M_k31 --reads 50000000 --cpu-sample 0 --extras 0
run C5_400M_k31_d1_sigma2.5 --reads 400000000 --sigma 2.5 --d 1 --cpu-sample 0 --extras 0
timeout 700 python tools/e2e_pregraph.py --reads 200000000 --p 16 --T 20000 --layout pe --skip-ref --timeout 200 --runs 3 --pause 20 > $O/e2e_pregraph_200M_k31_p16_pe_ours_only.json 2> $O/e2e_200M.err
timeout 1200 python tools/e2e_pregraph.py --reads 20000000 --p 16 --T 20000 --layout pe --timeout 600 --runs 2 --also-cli-args "--gpus 4 --share-device" > $O/e2e_pregraph_20M_k31_p16_pe.json 2> $O/e2e_20M.err
python3 - $O <<'E'
import json, sys
o = sys.argv[1]
for f in ("e2e_pregraph_200M_k31_p16_pe_ours_only.json", "e2e_pregraph_20M_k31_p16_pe.json"):
    try:
        d = json.load(open(o + "/" + f))
        print(f, d.get("ours_walls_s"), d.get("ref_wall_s"), d.get("identical"), (d.get("also") or {}).get("wall_s"), (d.get("also") or {}).get("same_as_first_run"))
    except Exception as e:
        print(f, "FAILED", e)
E
