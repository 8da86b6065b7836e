"""The 3x3 layers in 'fp32x3' (three bf16 planes, six products) and 'fp32h2' (two fp16 planes, three products): forward, data gradient,
weight gradient, microseconds per launch.  usage: proto_bench.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


for (name, C, H) in (("layer1", 64, 56), ("layer2", 128, 28), ("layer3", 256, 14), ("layer4", 512, 7)):
    x = torch.relu(torch.randn(N, H, H, C, device=dev))
    dy = torch.randn(N, H, H, C, device=dev) * 1e-3
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    res = {}
    for mode in ("fp32x3", "fp32h2"):
        ops.set_conv_compute_dtype(mode)
        if mode == "fp32h2":
            ops.amax_compute(x)
            ops.amax_compute(dy)
        wp, wd = ops._pack_weights(w, C, 0), ops._pack_weights(w, C, 1)
        res[mode] = (timed(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)), timed(lambda: ops.conv_bwd_data(dy, wd, x.shape, C, 3, 3, 1, 1)),
                     timed(lambda: ops.conv_bwd_weight(x, dy, w.shape, 1, 1)))
    fl = 2.0 * N * H * H * C * C * 9
    a, b = res["fp32x3"], res["fp32h2"]
    print("%s %d frames  fwd %.0f -> %.0f us (x%.2f, %.0f TFLOP/s)  dgrad %.0f -> %.0f (x%.2f)  wgrad %.0f -> %.0f (x%.2f, %.0f TFLOP/s)" % (
        name, N, a[0], b[0], a[0] / b[0], fl / b[0] / 1e6, a[1], b[1], a[1] / b[1], a[2], b[2], a[2] / b[2], fl / b[2] / 1e6))
ops.set_conv_compute_dtype("fp32x3")
