#!/bin/bash
# same-box A/B: merged trunk call (default) vs the two-call two-stream schedule (LMKD_MERGE=0), alternating, 6 s between processes
out=${1:-gpurun_out/ab_merge}
mkdir -p $out
for dt in f32 bf16; do
  for r in 1 2; do
    for m in 1 0; do
      LMKD_MERGE=$m python bench.py --dtype $dt --steps 32 --warmup 5 --no-cpu-baseline --no-other-modes > $out/${dt}_m${m}_r${r}.json 2> $out/${dt}_m${m}_r${r}.err
      python - $out/${dt}_m${m}_r${r}.json $dt $m <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dtype %s merge %s: %.2f episodes/s  (repeat %.2f)  roofline frac %.3f  host enqueue %.2f ms" % (sys.argv[2], sys.argv[3], d["value"], d["repeat"]["value"], d["roofline"]["frac"], d.get("host_enqueue_ms_per_episode", -1)))
P
      sleep 6
    done
  done
done
