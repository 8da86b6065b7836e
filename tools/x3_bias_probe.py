"""Is the error of the bf16-plane convolutions against fp64 BIASED (a non-zero mean, or a mean correlated with the sign of the
result)?  A zero-mean error of relative size 5e-7 per element averages out in the per-channel sums of the BatchNorm backward
(sum g, sum g*xhat over up to 2.5 M pixels); a directional one adds up coherently and shows as an error of those sums that
grows like sqrt(pixels) relative to a zero-mean implementation.
usage: python tools/x3_bias_probe.py [frames]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import litemkd_amd  # noqa: F401
from litemkd_amd import ops

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.set_num_threads(16)


def stats(name, y, ref):
    y, ref = y.double().cpu(), ref.double()
    e = y - ref
    rms = float(ref.pow(2).mean().sqrt())
    print("  %-34s rms err %.3e   mean err %+.3e   mean err*sign(ref) %+.3e   (all / rms of the result %.3f; %d elements -> noise floor of a mean %.1e)"
          % (name, float(e.pow(2).mean().sqrt()) / rms, float(e.mean()) / rms, float((e * ref.sign()).mean()) / rms, rms, e.numel(),
             float(e.pow(2).mean().sqrt()) / rms / e.numel() ** 0.5))


for (H, C, relu_in) in ((28, 128, True), (28, 128, False), (14, 256, True), (56, 64, True)):
    g = torch.Generator(device=dev).manual_seed(H * 1000 + C)
    x = torch.randn(N, H, H, C, device=dev, generator=g)
    if relu_in:
        x = torch.relu(x)
    w = torch.randn(C, C, 3, 3, device=dev, generator=g) * (2.0 / (C * 9)) ** 0.5
    sub = list(range(4)) + list(range(N - 4, N))      # both halves of the launch's row tiles (the second half accumulates -y)
    x64 = x[sub].permute(0, 3, 1, 2).cpu().double()
    ref = F.conv2d(x64, w.cpu().double(), None, 1, 1)
    cpu32 = F.conv2d(x64.float(), w.cpu(), None, 1, 1)
    print("3x3 conv %d ch %dx%d, %d frames, input %s:" % (C, H, H, N, "relu(randn)" if relu_in else "randn"))
    stats("torch CPU fp32", cpu32, ref)
    for mode in ("fp32", "fp32x3", "fp32x3_9", "bf16"):
        ops.set_conv_compute_dtype(mode)
        y, _ = ops.conv_fwd(x, ops.pack_weights(w, C, 0), C, 3, 3, 1, 1, True)
        r = ref
        if mode == "bf16":      # the exact result of the rounded operands
            r = F.conv2d(x64.to(torch.bfloat16).double(), w.cpu().to(torch.bfloat16).double(), None, 1, 1)
        stats("HIP " + mode, y[sub].permute(0, 3, 1, 2), r)
ops.reset_compute_dtypes()
