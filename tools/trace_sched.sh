#!/bin/bash
# kernel trace of the benchmark's default schedule (merged + pipelined): per-queue busy time, gaps, per-kernel totals over ~25 episodes
out=${1:-gpurun_out/trace_sched}
dt=${2:-f32}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
LMKD_TIMED_EVENTS=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/t -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $dt --steps 48 --warmup 4 --no-cpu-baseline --no-other-modes --roofline-episodes 1 > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
f=$(find $GRAFT_REPO_ROOT/$out/t -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $f 0.15 0.4
python3 $GRAFT_REPO_ROOT/tools/overlap_trace.py $f 0.15 0.4 | head -40
rm -rf $GRAFT_REPO_ROOT/$out/t
