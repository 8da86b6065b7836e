#!/bin/bash
# same-box A/B: two-call schedule | merged | merged + cross-episode pipelining | two-call + pipelining (alternating, 6 s apart)
out=${1:-gpurun_out/ab_pipe}
mkdir -p $out
for dt in f32 bf16; do
  for r in 1 2; do
    for cfg in "0 --no-pipeline" "1 --no-pipeline" "1 --pipeline" "0 --pipeline"; do
      set -- $cfg
      LMKD_MERGE=$1 python bench.py --dtype $dt --steps 32 --warmup 6 --no-cpu-baseline --no-other-modes $2 > $out/${dt}_m$1$2_r${r}.json 2> $out/${dt}_m$1$2_r${r}.err
      python - $out/${dt}_m$1$2_r${r}.json $dt "$cfg" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dtype %s merge/pipeline %-16s: %.2f episodes/s  (repeat %.2f)  host enqueue %.2f ms" % (sys.argv[2], sys.argv[3], d["value"], d["repeat"]["value"], d.get("host_enqueue_ms_per_episode", -1)))
P
      sleep 6
    done
  done
done
