#!/bin/bash
# the round's measurements on one box: tests, bench lines, rocprofv3 kernel statistics and PMC passes of the timed config
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_final
rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 1800 python bench.py > $O/bench_default_200M_k31.json 2> $O/bench_default_200M_k31.err
tail -1 $O/bench_default_200M_k31.json | cut -c1-250
bash tools/pmc_pipeline.sh $O/pmc200
python3 tools/pmc_pipeline_summary.py $O/pmc200 200000000 150 31 1 $O/pmc_pass1_200M_k31.json
find $O/pmc200 -name "pass_*" -type d | xargs rm -rf
timeout 900 python bench.py --reads 50000000 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_C2_50M_k31.json 2> $O/bench_C2.err
tail -1 $O/bench_C2_50M_k31.json | cut -c1-200
timeout 900 python bench.py --reads 50000000 --read-len 250 --K 63 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_C4_50M_250bp_k63.json 2> $O/bench_C4.err
tail -1 $O/bench_C4_50M_250bp_k63.json | cut -c1-200
timeout 1200 python bench.py --reads 400000000 --sigma 2.5 --d 1 --steps 2 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_C5_400M_k31_d1_sigma2.5.json 2> $O/bench_C5.err
tail -1 $O/bench_C5_400M_k31_d1_sigma2.5.json | cut -c1-200
timeout 900 python bench.py --pipeline direct --steps 2 --warmup 1 --cpu-sample 0 --extras 0 > $O/bench_direct_200M_k31.json 2> $O/bench_direct.err
tail -1 $O/bench_direct_200M_k31.json | cut -c1-200
for n in 2 4; do
SDT_SHM_OUTBOX_MB=6000 SDT_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2971$n bench.py --gpus $n --reads 40000000 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_${n}ranks_sharing_one_gpu_40M.json 2> $O/bench_share_$n.err
tail -1 $O/bench_${n}ranks_sharing_one_gpu_40M.json | cut -c1-200
done
timeout 1500 python tools/e2e_pregraph.py --reads 200000000 --read-len 150 --K 31 --p 16 --T 20000 --skip-ref --timeout 1200 > $O/e2e_pregraph_200M_k31_p16_ours_only.json 2> $O/e2e_200M.err
grep -E "ours_wall_s|ours_hash_only" $O/e2e_pregraph_200M_k31_p16_ours_only.json
