B="python bench.py --steps 24 --warmup 5 --no-cpu-baseline --no-other-modes"
r() { echo -n "$1: "; shift; env "$@" 2>/dev/null | cut -c60-95; }
for i in 1 2; do
r "A base          " X=1 $B
r "B main hi       " LMKD_PRIO=-1,-1,0 LMKD_MAIN_LANE=1 $B
r "B2 wgrad hi     " LMKD_PRIO=0,0,-1 LMKD_MAIN_LANE=1 $B
r "C pipe          " X=1 $B --pipeline
r "D pipe l0hi     " LMKD_PRIO=-1,0,0 $B --pipeline
r "E pipe lanes hi " LMKD_PRIO=-1,-1,0 $B --pipeline
r "F pipe wgrad hi " LMKD_PRIO=0,0,-1 $B --pipeline
done
