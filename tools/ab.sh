#!/bin/bash
# Same-box A/B of two builds of the library (device-to-device clocks differ by up to 12 %, so two gpurun calls cannot be compared):
# A = lite-mkd_amd/liblmkd_hip.so, B = lite-mkd_amd/build/prev.so, run A B A B with the given bench arguments.
# usage (GPU box, repo root): tools/ab.sh [bench.py arguments]
L=lite-mkd_amd/liblmkd_hip.so
cp $L /tmp/A.so; cp lite-mkd_amd/build/prev.so /tmp/B.so
for v in A B A B; do
  cp /tmp/$v.so $L
  python bench.py --steps 32 --no-cpu-baseline --no-other-modes --roofline-episodes 0 "$@" > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
  echo "$v $(tail -1 /tmp/ab.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["dtype"], round(d["value"],2), "episodes/s")')"
done
cp /tmp/A.so $L
