"""fp32-as-3xbf16 conv kernels vs the native fp32 MFMA kernels: speed per ResNet-18 layer shape (200 frames) and error of
both against an fp64 convolution (8 frames).  `gpurun -- python tools/conv_bench_x3.py`.  Tuning aid."""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
from litemkd_amd._lib import lib
dev = torch.device("cuda", 0)
def tm(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
MODES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp32", "fp32x3", "fp32x3_9"]
TILES = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for (Cin, H, Cout, K, s, p) in [(64,56,64,3,1,1),(64,56,128,3,2,1),(128,28,128,3,1,1),(128,28,256,3,2,1),(256,14,256,3,1,1),(256,14,512,3,2,1),(512,7,512,3,1,1),(64,56,128,1,2,0)]:
    torch.manual_seed(0)
    Ho = (H + 2 * p - K) // s + 1
    w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
    xs = torch.relu(torch.randn(8, H, H, Cin, device=dev)); gys = torch.randn(8, Ho, Ho, Cout, device=dev)
    yref = F.conv2d(xs.permute(0, 3, 1, 2).double(), w.double(), stride=s, padding=p)
    wref = torch.nn.grad.conv2d_weight(xs.permute(0, 3, 1, 2).double(), tuple(w.shape), gys.permute(0, 3, 1, 2).double(), stride=s, padding=p)
    dref = torch.nn.grad.conv2d_input((8, Cin, H, H), w.double(), gys.permute(0, 3, 1, 2).double(), stride=s, padding=p)
    x = torch.relu(torch.randn(200, H, H, Cin, device=dev)); gy = torch.randn(200, Ho, Ho, Cout, device=dev)
    fl = 2.0 * 200 * Ho * Ho * Cout * Cin * K * K
    line = "conv Cin%3d H%2d Cout%3d k%d s%d:" % (Cin, H, Cout, K, s)
    for mode in MODES:
        ops.set_conv_compute_dtype(mode)
        wp, wd = ops._pack_weights(w, Cin, 0), ops._pack_weights(w, Cin, 1)
        ys = ops.conv_fwd(xs, wp, Cout, K, K, s, p, True)[0].permute(0, 3, 1, 2).double()
        ds = ops.conv_bwd_data(gys, wd, (8, H, H, Cin), Cout, K, K, s, p).permute(0, 3, 1, 2).double()
        ey = ((ys - yref).norm() / yref.norm()).item(); ed = ((ds - dref).norm() / dref.norm()).item()
        line += "  [%s err %.2e %.2e" % (mode, ey, ed)
        for tile in (TILES if mode != "fp32" else [0]):
            lib().call("lmkd_conv_set_tile", tile)
            t1 = tm(lambda: ops.conv_fwd(x, wp, Cout, K, K, s, p, True))
            t2 = tm(lambda: ops.conv_bwd_data(gy, wd, (200, H, H, Cin), Cout, K, K, s, p))
            line += " | t%d fwd %5.1f dgrad %5.1f" % (tile, fl/t1/1e9, fl/t2/1e9)
        t3 = tm(lambda: ops.conv_bwd_weight(x, gy, tuple(w.shape), s, p))
        dws = ops.conv_bwd_weight(xs, gys, tuple(w.shape), s, p).double()
        line += " | wgrad %5.1f err %.2e" % (fl/t3/1e9, ((dws - wref).norm() / wref.norm()).item())
        lib().call("lmkd_conv_set_tile", 0)
        line += "]"
    ops.set_conv_compute_dtype("fp32")
    print(line, flush=True)
