#!/bin/bash
# kernel trace of one whole sdt-pregraph run (rocprofv3 wants the program itself after --: the FASTQ files are written first, by
# tools/e2e_pregraph.py --gen-only).  usage: tools/e2e_profile.sh <reads> [layout] [out dir]     (on the GPU box, from the repo root)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
READS=${1:-200000000}; LAYOUT=${2:-pe}; O=${3:-gpurun_out/e2e_prof}; D=/tmp/sdt_e2e_prof
mkdir -p $O; rm -rf $D
CMD=$(python tools/e2e_pregraph.py --reads $READS --p 16 --T 20000 --layout $LAYOUT --gen-only $D | tail -1)
cat $D/*.fq > /dev/null
echo "$CMD" > $O/cmd.txt
for i in 1; do t0=$(date +%s.%N); SDT_TIMING=1 $CMD > $O/run$i.out 2> $O/run$i.err; t1=$(date +%s.%N); echo "run $i: wall $(python3 -c "print(round($t1 - $t0, 2))") s; $(tail -1 $O/run$i.err)"; done
# (SDT_SLOW_EXIT: the CLI leaves through _exit otherwise, and the profiler writes its files from an exit handler)
ROOT=$(pwd)
(cd /tmp && SDT_SLOW_EXIT=1 SDT_TIMING=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -o e2e -- $CMD > $ROOT/$O/prof.out 2> $ROOT/$O/prof.err)
python3 - $O <<'E'
import csv, glob, sys
o = sys.argv[1]
for f in glob.glob(o + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(o + "/kernel_stats_e2e.csv", "w") as fo:
        fo.write("kernel,calls,total_ms,avg_ms,pct\n")
        for r in rows:
            fo.write('"%s",%s,%.2f,%.3f,%s\n' % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
    for r in rows[:45]:
        print("%-90s %6s %9.1f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
E
find $O/trace -name "*kernel_trace.csv" -size +20M -delete
rm -rf $D
