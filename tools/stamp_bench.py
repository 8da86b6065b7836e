"""phase times of the layer-1 two-plane patch kernel (conv_patch_x3_kernel) from s_memtime stamps of every 64th workgroup
(a -DLMKD_STAMPS build: tools/ab_build.sh stamps -DLMKD_STAMPS).   usage: LMKD_LIB=.../liblmkd_stamps.so python tools/stamp_bench.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
L = int(os.environ.get("LAYER", "1"))
C, H, N = {1: (64, 56, 400), 2: (128, 28, 400), 3: (256, 14, 400), 4: (512, 7, 400)}[L]
if os.environ.get("PATCH16"):
    litemkd_amd.lib().call("lmkd_conv_set_patch16", int(os.environ["PATCH16"]))
cd = ctypes.CDLL(os.environ["LMKD_LIB"])
ops.set_conv_compute_dtype("fp32h2")
x = ops.amax_compute(torch.relu(torch.randn(N, H, H, C, device=dev)))
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
wp = ops._pack_weights(w, C, 0)
for _ in range(3):
    ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)
torch.cuda.synchronize()
cd.lmkd_debug_stamps(None, 1)
ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
cd.lmkd_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
t = buf.reshape(4096, 8).astype(np.int64)
t = t[t[:, 4] > 0]
d = np.diff(t[:, :5], axis=1)
print("%d workgroups sampled; cycles (mean | median): prologue %.0f | %.0f, first patch + fragments arrive %.0f | %.0f, K loop %.0f | %.0f, epilogue %.0f | %.0f, life %.0f" % (
    len(t), d[:, 0].mean(), np.median(d[:, 0]), d[:, 1].mean(), np.median(d[:, 1]), d[:, 2].mean(), np.median(d[:, 2]), d[:, 3].mean(), np.median(d[:, 3]),
    (t[:, 4] - t[:, 0]).mean()))

