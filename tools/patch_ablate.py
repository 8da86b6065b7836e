"""What does conv_patch_x3_kernel wait for?  Timing ablations (lmkd_conv_set_patch_debug: parts of the work left out, results garbage)
of the benchmark's main tile (128x64, four waves, three workgroups per CU) on the four 3x3 layers at 200 frames, headline arithmetic.
usage (GPU box): python tools/patch_ablate.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
from litemkd_amd._lib import lib
dev = torch.device("cuda", 0)


def tm(f, reps=8):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N = 200
ops.set_conv_compute_dtype("fp32x3")
lib().call("lmkd_conv_set_tile", 11)
lib().call("lmkd_conv_set_patch16", 0)      # the ablation instances are variants of the 32x32x16 kernel
NAMES = {0: "full", 1: "-B loads", 2: "-A LDS reads", 4: "-split/store", 8: "-out stores", 16: "16B out stores", 15: "MFMA only", 32: "16x16x32 MFMAs", 47: "16x16x32 only"}
print("%-4s" % "", " ".join("%14s" % NAMES[d] for d in NAMES))
for (name, C, H) in (("l1", 64, 56), ("l2", 128, 28), ("l3", 256, 14), ("l4", 512, 7)):
    x = torch.relu(torch.randn(N, H, H, C, device=dev))
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wp = ops._pack_weights(w, C, 0)
    fl = 2.0 * N * H * H * C * C * 9
    best = {d: 1e9 for d in NAMES}
    for rnd_ in range(3):
        for d in NAMES:
            lib().call("lmkd_conv_set_patch_debug", d)
            best[d] = min(best[d], tm(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)))
    lib().call("lmkd_conv_set_patch_debug", 0)
    lib().call("lmkd_conv_set_patch16", 1)
    real16 = min(tm(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)) for _ in range(3))
    lib().call("lmkd_conv_set_patch16", 0)
    print("%-4s conv_patch16_x3_kernel (the product kernel): %6.1f us %4.0fTF" % (name, real16 * 1e3, fl / real16 / 1e9))
    print("%-4s" % name, " ".join("%6.1f us %4.0fTF" % (best[d] * 1e3, fl / best[d] / 1e9) for d in NAMES), flush=True)
lib().call("lmkd_conv_set_tile", 0)
