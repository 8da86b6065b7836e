#!/bin/bash
# same-box A/B of two builds of liblmkd_hip.so with the same header: ab_prev_lib.so (repo root, built from the previous sources) against
# the in-tree library.  usage (GPU box, repo root): tools/ab_lib.sh [bench flags]
cp lite-mkd_amd/liblmkd_hip.so /tmp/new.so
for i in 1 2 3; do
  sleep 6; cp ab_prev_lib.so lite-mkd_amd/liblmkd_hip.so; echo -n "prev: "; python bench.py --steps 24 --warmup 5 --no-cpu-baseline --no-other-modes "$@" 2>/dev/null | cut -c60-95
  sleep 6; cp /tmp/new.so lite-mkd_amd/liblmkd_hip.so; echo -n "new:  "; python bench.py --steps 24 --warmup 5 --no-cpu-baseline --no-other-modes "$@" 2>/dev/null | cut -c60-95
done
