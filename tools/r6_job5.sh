#!/bin/bash
# round 6, fifth GPU call: where the wall clock of sdt-pregraph goes at the bench's e2e size (8 M reads), at 20 M reads with 4 ranks
# sharing the device (the reader that parses each chunk once), and at the headline size (200 M paired-end reads)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_job5
mkdir -p $O
timeout 600 python3 tools/e2e_pregraph.py --reads 8000000 --layout se --K 31 --p 16 --T 20000 --skip-ref --runs 2 > $O/e2e_8M_se_ours_only.json 2> $O/e2e_8M.err
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r6_job5/e2e_8M_se_ours_only.json"))
print(j.get("ours_walls_s")); print("\n".join(j.get("ours_phase_ms", [])))
PY
timeout 900 python3 tools/e2e_pregraph.py --reads 20000000 --layout pe --K 31 --p 16 --T 20000 --skip-ref --also-cli-args "--gpus 4 --share-device" > $O/e2e_20M_pe_gpus4_shared.json 2> $O/e2e_20M.err
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r6_job5/e2e_20M_pe_gpus4_shared.json"))
print(j.get("ours_walls_s")); print(json.dumps(j.get("also"), indent=1)[:3000])
PY
timeout 1500 python3 tools/e2e_pregraph.py --reads 200000000 --layout pe --K 31 --p 16 --T 20000 --skip-ref --runs 2 --pause 20 > $O/e2e_200M_pe_ours_only.json 2> $O/e2e_200M.err
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r6_job5/e2e_200M_pe_ours_only.json"))
print(j.get("ours_walls_s"), j.get("gen_s")); print("\n".join(j.get("ours_phase_ms", [])))
PY
