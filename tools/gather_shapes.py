import sys, os, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
from litemkd_amd._lib import lib
dev = torch.device("cuda", 0)
def tm(f, reps=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 200
for mode in ("fp32x3", "bf16"):
    ops.set_conv_compute_dtype(mode)
    x = torch.randn(N, 224, 224, 4, device=dev); x[..., 3] = 0
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
    wp = ops._pack_weights(w, 4, 0)
    fl = 2.0 * N * 112 * 112 * 64 * 147
    gy = torch.randn(N, 112, 112, 64, device=dev)
    best = {}
    for r in range(3):
        for t in ("gather", "rows"):
            lib().call("lmkd_conv_set_stem_patch", int(t == "rows"))
            best[t] = min(best.get(t, 1e9), tm(lambda: ops.conv_fwd(x, wp, 64, 7, 7, 2, 3, True)))
    lib().call("lmkd_conv_set_stem_patch", 1)
    tw = tm(lambda: ops.conv_bwd_weight(x, gy, (64, 3, 7, 7), 2, 3))
    print(mode, {k: "%.0f us (%.0f TF)" % (v * 1e3, fl / v / 1e9) for k, v in best.items()}, "wgrad %.0f us (%.0f TF)" % (tw * 1e3, fl / tw / 1e9))
    # stride-2 3x3 and downsample shapes
    for (name, C, H, Cout, K, s, p) in [("l2.0.c1", 64, 56, 128, 3, 2, 1), ("l3.0.c1", 128, 28, 256, 3, 2, 1), ("l4.0.c1", 256, 14, 512, 3, 2, 1), ("l2.ds", 64, 56, 128, 1, 2, 0)]:
        xx = torch.randn(N, H, H, C, device=dev); ww = torch.randn(Cout, C, K, K, device=dev) * 0.05
        Ho = (H + 2 * p - K) // s + 1
        g2 = torch.randn(N, Ho, Ho, Cout, device=dev)
        wp2, wd2 = ops._pack_weights(ww, C, 0), ops._pack_weights(ww, C, 1)
        f2 = 2.0 * N * Ho * Ho * Cout * C * K * K
        t1 = tm(lambda: ops.conv_fwd(xx, wp2, Cout, K, K, s, p, True)); t2 = tm(lambda: ops.conv_bwd_data(g2, wd2, (N, H, H, C), Cout, K, K, s, p)); t3 = tm(lambda: ops.conv_bwd_weight(xx, g2, (Cout, C, K, K), s, p))
        print("   %-8s fwd %5.0f us (%5.1f TF) dgrad %5.0f us (%5.1f) wgrad %5.0f us (%5.1f)" % (name, t1 * 1e3, f2 / t1 / 1e9, t2 * 1e3, f2 / t2 / 1e9, t3 * 1e3, f2 / t3 / 1e9))
