for i in 1 2; do for v in True False; do python -c "
import sys; sys.argv=['bench.py','--steps','32','--no-cpu-baseline','--no-other-modes','--roofline-episodes','0'] + sys.argv[1:]
import litemkd_amd.ops as o; o.FUSE_PRE_ALL_MODES=$v
import bench; bench.main()" "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('PRE=$v', d['dtype'], round(d['value'],2))"; done; done
