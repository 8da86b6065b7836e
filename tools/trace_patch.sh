#!/bin/bash
# true kernel durations (rocprofv3 --kernel-trace --stats) of tools/patch_bench.py in one mode: the python timing loop of that tool is
# host-bound for kernels shorter than ~70 us
mode=${1:-bf16act}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace_patch_$mode
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o r -- python3 $R/tools/patch_bench.py $mode > $out/log.txt 2>&1
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python3 - $f <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%8.1f us avg  %6d calls  min %7.1f max %7.1f  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:120]))
P
rm -rf $out/t
