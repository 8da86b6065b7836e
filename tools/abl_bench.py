"""fp32h2 3x3 layers, forward (BatchNorm sums on) / data gradient / weight gradient, microseconds per launch, with the patch kernel on
v_mfma_f32_16x16x32_f16 (conv_patch16_x3_kernel) and on v_mfma_f32_32x32x16_f16 (conv_patch_x3_kernel: lmkd_conv_set_patch16(0)).
usage: abl_bench.py [frames] [layers e.g. 1,3]   (LMKD_LIB: another build, tools/ab_build.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
which = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
L = litemkd_amd.lib()


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
for (li, C, H) in ((1, 64, 56), (2, 128, 28), (3, 256, 14), (4, 512, 7)):
    if li not in which:
        continue
    x = torch.relu(torch.randn(N, H, H, C, device=dev))
    dy = torch.randn(N, H, H, C, device=dev) * 1e-3
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    ops.amax_compute(x)
    ops.amax_compute(dy)
    wp, wd = ops._pack_weights(w, C, 0), ops._pack_weights(w, C, 1)
    out = []
    ys = {}
    for p16 in (2, 0):
        L.call("lmkd_conv_set_patch16", p16)
        tf = timed(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True))
        tn = timed(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, False))
        td = timed(lambda: ops.conv_bwd_data(dy, wd, x.shape, C, 3, 3, 1, 1))
        ys[p16] = ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)[0]
        out.append("%s fwd+stats %.0f fwd %.0f dgrad %.0f" % ("16x16x32" if p16 else "32x32x16", tf, tn, td))
    L.call("lmkd_conv_set_patch16", 1)
    tw = timed(lambda: ops.conv_bwd_weight(x, dy, w.shape, 1, 1))
    print("L%d" % li, " | ".join(out), "| wgrad %.0f | max diff between the two %.2e of %.2e" % (tw, float((ys[0] - ys[2]).abs().max()), float(ys[2].abs().max())))
