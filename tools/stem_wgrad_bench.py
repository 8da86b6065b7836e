"""the stem's weight gradient alone (200 frames of 224 x 224): stem_wgrad_kernel vs the im2col-gather kernel, microseconds per call"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import litemkd_amd  # noqa: F401
from litemkd_amd import ops

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.zeros(N, 224, 224, 4, device=dev)
x[..., :3] = torch.rand(N, 224, 224, 3, device=dev)
dy = torch.randn(N, 112, 112, 64, device=dev) * 1e-3
for on in (0, 1):
    ops.lib().call("lmkd_conv_set_wgrad_stem", on)
    f = lambda: ops.conv_bwd_weight(x, dy, (64, 3, 7, 7), 2, 3)
    for _ in range(3):
        dw = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("stem_wgrad_kernel" if on else "gather kernel    ", "%.1f us  %.1f TFLOP/s" % (us, 2.0 * N * 112 * 112 * 64 * 147 / us / 1e6), float(dw.abs().sum()))
