#!/bin/bash
# round 6, sixteenth GPU call: the SQ-counter passes again, for the kernels as they are at the end of the round
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job16
mkdir -p $O
timeout 1500 bash tools/pmc_sq.sh $O/sq --est-distinct 809675638
python3 tools/pmc_sq_summary.py $O/sq 24000000000 $O/sq_pass1_200M_k31_final.json > $O/sq_summary_final.txt 2>&1
cat $O/sq_summary_final.txt | cut -c1-900
find $O/sq -name "*.csv" -size +1M -delete
