cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r3_sq/counters_list.txt 2>&1 | true
bash tools/pmc_sq.sh gpurun_out/r3_sq/base
python3 tools/pmc_sq_summary.py gpurun_out/r3_sq/base 24000000000 gpurun_out/r3_sq/sq_pass1_200M_k31_r2build.json
find gpurun_out/r3_sq -name "pass_*" -type d | xargs rm -rf
