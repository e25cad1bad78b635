#!/bin/bash
# round 6, fourteenth GPU call: BASELINE config 5 end to end at 20 M reads (both pipelines, the reference's contig / scaff on our output),
# and the kernel trace of one whole sdt-pregraph run at 200 M paired-end reads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job14
mkdir -p $O
timeout 2400 python3 tools/e2e_pipeline.py --reads 20000000 --sigma 2.5 --d 1 --p 16 > $O/e2e_pipeline_20M_sigma2.5_d1.json 2> $O/e2e_pipeline.err
cat $O/e2e_pipeline_20M_sigma2.5_d1.json | head -60
timeout 900 bash tools/e2e_profile.sh 200000000 pe $O/e2e_prof > $O/e2e_prof.log 2>&1; tail -30 $O/e2e_prof.log | cut -c1-130
