#!/bin/bash
# same-box A/B of the pipeline depth (forwards queued ahead of the oldest pending backward): 1 vs 2 (alternating, 6 s apart)
out=${1:-gpurun_out/ab_depth}
mkdir -p $out
for dt in f32 bf16; do
  for r in 1 2; do
    for d in 1 2; do
      LMKD_PIPE_DEPTH=$d python bench.py --dtype $dt --steps 32 --warmup 6 --no-cpu-baseline --no-other-modes > $out/${dt}_d${d}_r${r}.json 2> $out/${dt}_d${d}_r${r}.err
      python - $out/${dt}_d${d}_r${r}.json $dt $d <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dtype %s pipeline depth %s: %.2f episodes/s  (repeat %.2f)  host enqueue %.2f ms" % (sys.argv[2], sys.argv[3], d["value"], d["repeat"]["value"], d.get("host_enqueue_ms_per_episode", -1)))
P
      sleep 6
    done
  done
done
