"""Error of the two-plane fp16 arithmetic ('fp32h2': fp32 = 2 x fp16 with power-of-two scales, three products) against fp64, beside the
three-plane bf16 arithmetic ('fp32x3', six products) and torch's fp32 convolution, on the four 3x3 shapes of the trunk: forward, data
gradient, weight gradient.  usage: h2_error.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)


def run(mode, x, w, dy):
    ops.set_conv_compute_dtype(mode)
    C = w.shape[0]
    for t in (x, dy):
        if hasattr(t, "_lmkd_amax"):
            del t._lmkd_amax
    if mode == "fp32h2":
        ops.amax_compute(x)
        ops.amax_compute(dy)
    n0 = lib.value("lmkd_conv_h2_launches")
    y = ops.conv_fwd(x, ops._pack_weights(w, C, 0), C, 3, 3, 1, 1, True)[0]
    dx = ops.conv_bwd_data(dy, ops._pack_weights(w, C, 1), x.shape, C, 3, 3, 1, 1)
    dw = ops.conv_bwd_weight(x, dy, w.shape, 1, 1)
    torch.cuda.synchronize()
    assert lib.value("lmkd_conv_h2_launches") - n0 == (3 if mode == "fp32h2" else 0), "which kernels ran"
    return y, dx, dw


print("%d frames; relative L2 error against fp64 (forward | data gradient | weight gradient)" % N)
for (name, C, H, kind) in (("layer1", 64, 56, "relu"), ("layer2", 128, 28, "relu"), ("layer3", 256, 14, "relu"), ("layer4", 512, 7, "relu"),
                           ("layer2, e^(3 N(0,1)) spread", 128, 28, "lognormal"), ("layer3, 1e-6 gradients", 256, 14, "grad")):
    if kind == "relu":
        x = torch.relu(torch.randn(N, H, H, C, device=dev))
        dy = torch.randn(N, H, H, C, device=dev) * 1e-3
    elif kind == "lognormal":
        x = torch.randn(N, H, H, C, device=dev) * torch.exp(3.0 * torch.randn(N, H, H, C, device=dev))
        dy = torch.randn(N, H, H, C, device=dev) * torch.exp(3.0 * torch.randn(N, H, H, C, device=dev))
    else:
        x = torch.relu(torch.randn(N, H, H, C, device=dev))
        dy = torch.randn(N, H, H, C, device=dev) * 1e-6 * torch.exp(torch.randn(1, 1, 1, C, device=dev) * 2)
    w = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    xd = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, padding=1)
    yd.backward(dy.permute(0, 3, 1, 2).double())
    ref = (yd.detach().permute(0, 2, 3, 1), xd.grad.permute(0, 2, 3, 1), wd.grad)
    xf = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wf = w.clone().requires_grad_(True)
    yf = F.conv2d(xf, wf, padding=1)
    yf.backward(dy.permute(0, 3, 1, 2))
    t32 = (yf.detach().permute(0, 2, 3, 1), xf.grad.permute(0, 2, 3, 1), wf.grad)
    rel = lambda got: " | ".join("%.2e" % (float((g.double() - r).norm()) / float(r.norm())) for g, r in zip(got, ref))
    print("%-30s C=%3d  torch fp32  %s" % (name, C, rel(t32)))
    print("%-30s        fp32x3      %s" % ("", rel(run("fp32x3", x, w, dy))))
    print("%-30s        fp32h2      %s" % ("", rel(run("fp32h2", x, w, dy))))
ops.set_conv_compute_dtype("fp32x3")
