#!/bin/bash
# kernel traces of the merged and the two-call schedule (16 episodes each) -> timeline_gaps / overlap_trace summaries
out=${1:-gpurun_out/trace_merge}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  LMKD_MERGE=$m LMKD_TIMED_EVENTS=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/m$m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-other-modes --roofline-episodes 1 > $GRAFT_REPO_ROOT/$out/m$m.json 2> $GRAFT_REPO_ROOT/$out/m$m.err
  f=$(find $GRAFT_REPO_ROOT/$out/m$m -name "*kernel_trace.csv" | head -1)
  echo "== LMKD_MERGE=$m  $f"
  python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 0.15 0.4
  python3 $GRAFT_REPO_ROOT/tools/overlap_trace.py $f 0.15 0.4 | head -24
  rm -rf $GRAFT_REPO_ROOT/$out/m$m
done
