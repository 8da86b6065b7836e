"""How much does the last partial round of workgroups cost the patch kernel?  Layer-4 shape (512 ch, 7x7, 3x3) at frame counts that
give 0.8 / 1.0 / 1.2 / 1.6 / 2.0 rounds of the 768 resident workgroups (128x64 tiles, three per CU): us per launch and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
dev = torch.device("cuda", 0)
C = 512
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
wp = ops._pack_weights(w, C, 0)
for N in (160, 200, 250, 300, 400, 500):
    x = torch.relu(torch.randn(N, 7, 7, C, device=dev))
    f = lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 100)
    tiles = -(-N * 49 // 128) * 8
    print("N %3d: %4d workgroups = %.2f rounds of 768   %.1f us   %.1f TFLOP/s" % (N, tiles, tiles / 768, best, 2.0 * N * 49 * C * C * 9 / best / 1e6))
