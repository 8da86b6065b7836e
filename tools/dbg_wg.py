import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
for (N, C, H, Cout, K, s, p) in [(80, 512, 2, 512, 3, 1, 1), (40, 512, 2, 512, 3, 1, 1), (80, 256, 4, 512, 3, 2, 1), (40, 128, 8, 128, 3, 1, 1), (7, 128, 5, 128, 3, 1, 1)]:
    torch.manual_seed(0)
    x = torch.relu(torch.randn(N, H, H, C, device=dev)); Ho = (H + 2 * p - K) // s + 1
    gy = torch.randn(N, Ho, Ho, Cout, device=dev)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double().cpu(), (Cout, C, K, K), gy.permute(0, 3, 1, 2).double().cpu(), stride=s, padding=p)
    line = "N%d C%d H%d Cout%d s%d:" % (N, C, H, Cout, s)
    for mode in ("fp32", "fp32x3", "bf16"):
        ops.set_conv_compute_dtype(mode)
        dw = ops.conv_bwd_weight(x, gy, (Cout, C, K, K), s, p).double().cpu()
        e = dw - ref
        line += "  %s %.2e (max abs %.2e at co %d)" % (mode, float(e.norm() / ref.norm()), float(e.abs().max()), int(e.abs().amax((1, 2, 3)).argmax()))
    ops.set_conv_compute_dtype("fp32")
    print(line)
