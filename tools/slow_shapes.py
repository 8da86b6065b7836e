"""Timing of the conv launches that sit furthest below the fp32 MFMA roofline (stem, stride-2 3x3, 1x1/2 downsample, layer-1
weight gradient) per tile configuration, 200 frames: `gpurun -- python tools/slow_shapes.py`.  Tuning aid."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
from litemkd_amd._lib import lib
dev = torch.device("cuda", 0)
def tm(f, reps=8):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 200
shapes = [("stem", 4, 3, 224, 64, 7, 2, 3), ("l2.0.conv1", 64, 64, 56, 128, 3, 2, 1), ("l3.0.conv1", 128, 128, 28, 256, 3, 2, 1),
          ("l4.0.conv1", 256, 256, 14, 512, 3, 2, 1), ("l2.ds", 64, 64, 56, 128, 1, 2, 0), ("l3.ds", 128, 128, 28, 256, 1, 2, 0),
          ("l4.ds", 256, 256, 14, 512, 1, 2, 0), ("l1.3x3", 64, 64, 56, 64, 3, 1, 1)]
only = sys.argv[1:] 
for (name, Cs, Cin, H, Cout, K, s, p) in shapes:
    if only and name not in only: continue
    x = torch.relu(torch.randn(N, H, H, Cs, device=dev)); w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
    Ho = (H + 2 * p - K) // s + 1
    gy = torch.randn(N, Ho, Ho, Cout, device=dev)
    wp = ops.pack_weights(w, Cs, 0)
    fl = 2.0 * N * Ho * Ho * Cout * Cin * K * K
    line = "%-11s" % name
    for cfg in (0, 3, 2, 6, 5):
        if Cout <= 64 and cfg == 5: continue
        lib().call("lmkd_conv_set_tile", cfg)
        t1 = tm(lambda: ops.conv_fwd(x, wp, Cout, K, K, s, p, True))
        line += "  [t%d fwd %5.1f" % (cfg, fl/t1/1e9)
        if Cin != 3:
            wd = ops.pack_weights(w, Cin, 1)
            t2 = tm(lambda: ops.conv_bwd_data(gy, wd, (N, H, H, Cin), Cout, K, K, s, p))
            line += " dgrad %5.1f" % (fl/t2/1e9)
        line += "]"
    lib().call("lmkd_conv_set_tile", 0)
    t3 = tm(lambda: ops.conv_bwd_weight(x, gy, tuple(w.shape), s, p))
    print(line + "  wgrad %5.1f TF (%.0f us)" % (fl/t3/1e9, t3 * 1e3), flush=True)
