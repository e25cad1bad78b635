run() { printf "%s %s: " "$1" "$2"; env $1 timeout 250 python bench.py --extras 0 --cpu-sample 0 $2 2>gpurun_out/x.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith(\"{\"):
        j=json.loads(l); r=j[\"roofline\"]; print(round(j[\"value\"]/1e9,2), round(j[\"ms_per_step\"],1), r[\"stage_ms_per_step\"], r[\"merges_per_kmer\"])
"; grep -h "table slots" gpurun_out/x.err; }
