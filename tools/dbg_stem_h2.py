"""diagnosis: the stem's weight gradient in fp32h2 against fp32x3 on realistic operands"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd
from litemkd_amd import ops
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (N, seg, H, scale_dy, kind) in ((8, 0, 224, 1.0, "randn"), (8, 0, 224, 1e-6, "randn"), (8, 3, 224, 1e-6, "img"), (400, 200, 224, 1e-6, "img"), (400, 0, 224, 1e-6, "img"), (8, 0, 64, 1e-6, "img")):
    x = torch.zeros(N, H, H, 4, device=dev)
    x[..., :3] = torch.rand(N, H, H, 3, device=dev) if kind == "img" else torch.randn(N, H, H, 3, device=dev)
    Ho = H // 2
    dy = torch.randn(N, Ho, Ho, 64, device=dev) * scale_dy
    dy[N // 2:] *= 0.05
    out = {}
    for mode in ("fp32x3", "fp32h2"):
        ops.set_conv_compute_dtype(mode)
        if mode == "fp32h2":
            ops.amax_compute(x, seg)
            ops.amax_compute(dy, seg)
        n0 = lib.value("lmkd_conv_h2_launches")
        out[mode] = ops.conv_bwd_weight(x, dy, (64, 3, 7, 7), 2, 3, seg=seg).clone()
        ran = lib.value("lmkd_conv_h2_launches") - n0
    a, b = out["fp32x3"], out["fp32h2"]
    print("N %3d seg %3d H %3d dy x %g %s: two-plane launches %d   |h2 - x3| / |x3| = %.3e   finite %s" % (
        N, seg, H, scale_dy, kind, ran, float((a - b).norm() / a.norm()), bool(torch.isfinite(b).all())))
ops.set_conv_compute_dtype("fp32x3")
