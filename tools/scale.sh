#!/bin/bash
# The scaling curve of the headline workload on ONE node, with what is needed to DIAGNOSE it, not just the number:
#   tools/scale.sh [reads] [steps]        (defaults: 200000000 reads, 3 steps; run where N GPUs are visible)
# For N = 1, 2, 4, 8 (as far as the node has GPUs): bench.py --gpus N under torch.distributed.run (one rank per GPU, the
# library's own RCCL data path), then per line: k-mers/s, efficiency against N x the single-GPU rate, skew of the k-mers the
# ranks counted into their shards (max / mean), bytes exchanged per k-mer, GB/s per xGMI link on rank 0, and rank 0's
# stage times.  SDT_COMM_TIMEOUT_S bounds every wait on a peer (a rank that leaves makes the others exit non-zero with the
# rank / round in the message instead of hanging).  Every run is under `timeout`.
cd "$(dirname "$0")/.." || exit 1
READS=${1:-200000000}; STEPS=${2:-3}
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 1)
export HSA_ENABLE_IPC_MODE_LEGACY=0 SDT_COMM_TIMEOUT_S=${SDT_COMM_TIMEOUT_S:-120}
OUT=${SCALE_OUT:-gpurun_out/scale}; mkdir -p "$OUT"
BASE=""
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && break
  LOG=$OUT/scale_n$N.json
  if [ "$N" = 1 ]; then
    timeout 900 python3 bench.py --gpus 1 --reads "$READS" --steps "$STEPS" --warmup 1 --cpu-sample 0 --extras 0 > "$LOG" 2> "$OUT/scale_n$N.err"
  else
    timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus "$N" --reads "$READS" --steps "$STEPS" --warmup 1 --cpu-sample 0 --extras 0 > "$LOG" 2> "$OUT/scale_n$N.err"
  fi
  RC=$?
  python3 - "$LOG" "$N" "$RC" "$BASE" <<'PY'
import json, sys
path, n, rc, base = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
line = next((l for l in open(path) if l.startswith("{")), None)
if line is None:
    print(f"N={n}: no result (exit code {rc}); see {path[:-5]}.err")
    sys.exit(0)
j = json.loads(line)
v = j["value"]
eff = f"{v / (n * float(base)):.2f}" if base else "1.00"
r = j.get("roofline") or {}
x = j.get("exchange") or {}
print(f"N={n}: {v / 1e9:7.2f} G k-mers/s  {j['ms_per_step']:8.1f} ms/step  efficiency {eff}  skew max/mean {j.get('skew_max_over_mean', 1.0)}"
      f"  exchange {x.get('bytes_per_kmer', 0)} B/k-mer, {x.get('GBps_per_link_rank0', 0)} GB/s per link (rank 0), {x.get('ms_on_exchange_stream_rank0', 0)} ms on the exchange stream"
      f"  rank-0 stages ms/step {r.get('stage_ms_per_step')}  per-rank k-mers {j.get('per_rank_kmers_counted')}")
PY
  [ "$N" = 1 ] && BASE=$(python3 -c "import json,sys; print(next(json.loads(l)['value'] for l in open('$LOG') if l.startswith('{')))" 2>/dev/null)
done
