"""Split plan of the 3x3 / stride-1 weight gradient (conv_wgrad_win16_kernel): time of one launch over two frame segments (the merged trunk call's
form: 200 + 200 frames) against the per-segment workgroup target (lmkd_conv_set_wgrad_window(n > 1)); the launch has 2 x target workgroups.
usage: python tools/wgrad_split_ab.py [targets ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
dev = torch.device("cuda", 0)
L = litemkd_amd.lib()
targets = [int(v) for v in sys.argv[1:]] or [512, 384, 768, 256, 192]


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


ops.set_conv_compute_dtype("fp32h2")
N, S = 400, 200
for (li, C, H) in ((1, 64, 56), (2, 128, 28), (3, 256, 14), (4, 512, 7)):
    x = torch.relu(torch.randn(N, H, H, C, device=dev))
    dy = torch.randn(N, H, H, C, device=dev) * 1e-3
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    ops.amax_compute(x, S)
    ops.amax_compute(dy, S)
    out = []
    for t in targets:
        L.call("lmkd_conv_set_wgrad_window", t)
        out.append("%d: %.0f" % (t, timed(lambda: ops.conv_bwd_weight(x, dy, w.shape, 1, 1, seg=S))))
    L.call("lmkd_conv_set_wgrad_window", 1)
    print("layer %d  two segments of 200 frames, us per launch by per-segment workgroup target:  %s" % (li, "   ".join(out)))
