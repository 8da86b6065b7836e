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

# two launches of the 200-frame shape on two streams at once (what the episode's two trunk calls do) against one 400-frame launch
print("--- two concurrent 200-frame launches (two streams) vs one 400-frame launch")
for Cc, Hh in ((512, 7), (256, 14), (128, 28), (64, 56)):
    ww = torch.randn(Cc, Cc, 3, 3, device=dev) * 0.02
    wpp = ops._pack_weights(ww, Cc, 0)
    xa, xb = torch.relu(torch.randn(200, Hh, Hh, Cc, device=dev)), torch.relu(torch.randn(200, Hh, Hh, Cc, device=dev))
    xab = torch.cat([xa, xb], 0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two():
        with torch.cuda.stream(s1):
            ops.conv_fwd(xa, wpp, Cc, 3, 3, 1, 1, True)
        with torch.cuda.stream(s2):
            ops.conv_fwd(xb, wpp, Cc, 3, 3, 1, 1, True)

    def one():
        ops.conv_fwd(xab, wpp, Cc, 3, 3, 1, 1, True)

    def single():
        ops.conv_fwd(xa, wpp, Cc, 3, 3, 1, 1, True)
    out = []
    for f in (single, two, one):
        torch.cuda.synchronize()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            e0.record()
            s1.wait_event(e0); s2.wait_event(e0)
            for _ in range(10):
                f()
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        out.append(best)
    print("%3d ch %2dx%2d: one 200-frame launch %.1f us | two at once %.1f us | one 400-frame launch %.1f us" % (Cc, Hh, Hh, out[0], out[1], out[2]))
