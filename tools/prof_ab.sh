#!/bin/bash
# per-kernel totals of the serialized benchmark (5 episodes) in THIS tree and in .ab_prev, same box: which kernels got slower?
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_ab; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
B="--warmup 1 --no-cpu-baseline --no-other-modes --serial --steps 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/new -o r -- python3 $R/bench.py $B --no-graph > $out/new.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prev -o r -- python3 $R/.ab_prev/bench.py $B > $out/prev.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
def load(d):
    f = sorted(glob.glob("gpurun_out/prof_ab/%s/**/*kernel_stats.csv" % d, recursive=True))[-1]
    t = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "")[:70]
        t[n] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 5e6)
    return t
a, b = load("new"), load("prev")
print("%-72s %8s %8s | %8s %8s" % ("kernel (ms per episode, serial)", "new n", "new ms", "prev n", "prev ms"))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1])):
    x, y = a.get(k, (0, 0.0)), b.get(k, (0, 0.0))
    if x[1] + y[1] > 0.02:
        print("%-72s %8.1f %8.3f | %8.1f %8.3f" % (k, x[0] / 5, x[1], y[0] / 5, y[1]))
print("total: new %.2f ms (%d launches/episode), prev %.2f ms (%d)" % (sum(v[1] for v in a.values()), sum(v[0] for v in a.values()) / 5, sum(v[1] for v in b.values()), sum(v[0] for v in b.values()) / 5))
PY
