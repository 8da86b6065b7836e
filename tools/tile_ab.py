"""3x3 layers in fp32h2 / fp32x3 with the patch tile pinned: 11 (128x64) against 12 (128x128) against the launcher's choice.  usage: tile_ab.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
lib = litemkd_amd.lib()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100


for mode in ("fp32h2", "fp32x3"):
    ops.set_conv_compute_dtype(mode)
    for (name, C, H) in (("layer1", 64, 56), ("layer2", 128, 28), ("layer3", 256, 14), ("layer4", 512, 7)):
        x = ops.amax_compute(torch.relu(torch.randn(N, H, H, C, device=dev)))
        dy = ops.amax_compute(torch.randn(N, H, H, C, device=dev) * 1e-3)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        wp, wd = ops._pack_weights(w, C, 0), ops._pack_weights(w, C, 1)
        out = []
        for tile in (0, 11, 12):
            if tile == 12 and C < 128:
                continue
            lib.call("lmkd_conv_set_tile", tile)
            out.append("tile %2d: fwd %4.0f dgrad %4.0f us" % (tile, timed(lambda: ops.conv_fwd(x, wp, C, 3, 3, 1, 1, True)),
                                                              timed(lambda: ops.conv_bwd_data(dy, wd, x.shape, C, 3, 3, 1, 1))))
        lib.call("lmkd_conv_set_tile", 0)
        print(mode, name, "   ".join(out))
ops.set_conv_compute_dtype("fp32x3")
