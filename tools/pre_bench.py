"""A/B: convolution with the BatchNorm+ReLU of its input fused into the loader vs bn_apply + plain convolution (200 frames)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
import sys
MODE = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ops.set_conv_compute_dtype(MODE)
for name, H, C, Cout in [("layer1", 56, 64, 64), ("layer2", 28, 128, 128), ("layer3", 14, 256, 256), ("layer4", 7, 512, 512)]:
    N = 200
    c1 = torch.randn(N, H, H, C, device=dev)
    st = torch.zeros(5, C, device=dev); st[1] = 1; st[2] = 1.0; st[3] = 0.1
    w = torch.randn(Cout, C, 3, 3, device=dev) * 0.05
    wp = ops.pack_weights(w, C, 0)
    a1 = ops.bn_apply(c1, st, True)
    dy = torch.randn(N, H, H, Cout, device=dev)
    fl = 2.0 * N * H * H * Cout * C * 9
    t_apply = timeit(lambda: ops.bn_apply(c1, st, True))
    t_f0 = timeit(lambda: ops.conv_fwd(a1, wp, Cout, 3, 3, 1, 1, True))
    t_f1 = timeit(lambda: ops.conv_fwd(c1, wp, Cout, 3, 3, 1, 1, True, pre_stats=st))
    t_w0 = timeit(lambda: ops.conv_bwd_weight(a1, dy, (Cout, C, 3, 3), 1, 1))
    t_w1 = timeit(lambda: ops.conv_bwd_weight(c1, dy, (Cout, C, 3, 3), 1, 1, pre_stats=st))
    print("%s bn_apply %.0f us | fwd plain %.0f us (%.1f TF) fused %.0f us (%.1f TF) | wgrad plain %.0f us (%.1f) fused %.0f us (%.1f)" % (
        name, t_apply, t_f0, fl / t_f0 / 1e6, t_f1, fl / t_f1 / 1e6, t_w0, fl / t_w0 / 1e6, t_w1, fl / t_w1 / 1e6))
