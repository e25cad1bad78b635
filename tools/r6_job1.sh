#!/bin/bash
# round 6, first GPU call: the judged bench line with the new e2e leg, the overlap probe, the SQ counter passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job1
mkdir -p $O
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
timeout 600 python3 tools/overlap_probe.py --masks > $O/overlap_probe.json 2> $O/overlap_probe.err
cat $O/overlap_probe.json; tail -5 $O/overlap_probe.err
timeout 1500 bash tools/pmc_sq.sh $O/sq
python3 tools/pmc_sq_summary.py $O/sq 24000000000 $O/sq_pass1_200M_k31.json > $O/sq_summary.txt 2>&1
cat $O/sq_summary.txt
find $O/sq -name "*.csv" -size +1M -delete
