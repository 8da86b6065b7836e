"""'fp32h2' against 'fp32x3' on whole training episodes (5-way 5-shot, 224^2): losses, flat gradient bucket, how many launches took
the two-plane form, milliseconds per episode.  usage: h2_episode.py [episodes] [serial|bench]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import litemkd_amd
from litemkd_amd.schedule import Schedule
from test_gpu_schedule import _run
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else "bench"
mk = Schedule.bench if which == "bench" else Schedule.serial
res = {}
for mode in ("fp32x3", "fp32h2", "fp32"):
    n0 = lib.value("lmkd_conv_h2_launches")
    _run(dev, mk(conv_dtype=mode), 5, 224, 1)      # warm
    torch.cuda.synchronize()
    t0 = time.time()
    r = _run(dev, mk(conv_dtype=mode), 5, 224, E)
    torch.cuda.synchronize()
    res[mode] = r
    print("%s: losses %s   two-plane launches %d   (%.1f ms per episode incl. set-up)" % (
        mode, [float("%.6f" % v) for v in r[0]], lib.value("lmkd_conv_h2_launches") - n0, (time.time() - t0) / E * 1e3))
for m1, m2 in (("fp32x3", "fp32h2"), ("fp32", "fp32x3"), ("fp32", "fp32h2")):
    a, b = res[m1], res[m2]
    print("%s vs %s: loss difference %.2e   gradient |d|max / |g|max = %.3e   rel-L2 = %.3e" % (
        m1, m2, (a[0] - b[0]).abs().max().item(), float((a[1] - b[1]).abs().max() / a[1].abs().max()), float((a[1] - b[1]).norm() / a[1].norm())))
