#!/bin/bash
# kernel trace + stats of the serialized benchmark in one arithmetic:  tools/trace_serial.sh <tag> [bench args]  -> gpurun_out/<tag>_stats.txt
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python3 $R/bench.py --warmup 1 --no-cpu-baseline --no-other-modes --serial --steps 4 "$@" > $out/log.txt 2>&1
cd $R
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/r_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/%s_stats.txt" % tag, "w") as w:
    w.write("total kernel time %.2f ms over the traced run\n" % (tot / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
        w.write("%6.2f%% %9.1f us avg x %5d  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:150]))
PY
