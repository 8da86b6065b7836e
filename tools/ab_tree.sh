#!/bin/bash
# Same-box A/B of two source TREES (kernels + Python): A = this tree, B = .ab_prev (a git worktree of an earlier commit with its own
# built liblmkd_hip.so).  Clocks differ by up to 12 % from device to device, so only alternating runs on one box compare.
# usage (GPU box, repo root): tools/ab_tree.sh [rounds] [-- bench.py arguments for both trees]
R=${1:-2}; shift; [ "$1" = "--" ] && shift
COMMON="--steps 32 --warmup 5 --no-cpu-baseline --no-other-modes --roofline-episodes 0 $@"
val() { tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["dtype"], round(d["value"],2), "episodes/s", "host", round(d.get("host_enqueue_ms_per_episode",-1),2))'; }
for i in $(seq $R); do
  echo "B prev       $(cd .ab_prev && python bench.py $COMMON 2>/dev/null | val)"
  echo "A new eager  $(python bench.py $COMMON --no-graph 2>/dev/null | val)"
  echo "A new graph  $(python bench.py $COMMON 2>/dev/null | val)"
done
