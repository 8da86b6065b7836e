import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
def tm(f, reps=8):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Cin = Cout = 64; H = 56; K = 3
w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
fl = 2.0 * 200 * H * H * Cout * Cin * 9
for mode in ("fp32", "fp32x3"):
    ops.set_conv_compute_dtype(mode)
    wp, wd = ops._pack_weights(w, Cin, 0), ops._pack_weights(w, Cin, 1)
    for name, x in (("dense randn", torch.randn(200, H, H, Cin, device=dev)), ("relu(randn)", torch.relu(torch.randn(200, H, H, Cin, device=dev))),
                    ("zeros", torch.zeros(200, H, H, Cin, device=dev)), ("small ints", torch.randint(0, 4, (200, H, H, Cin), device=dev).float())):
        t1 = tm(lambda: ops.conv_fwd(x, wp, Cout, K, K, 1, 1, True))
        t0 = tm(lambda: ops.conv_fwd(x, wp, Cout, K, K, 1, 1, False))
        t2 = tm(lambda: ops.conv_bwd_data(x, wd, (200, H, H, Cin), Cout, K, K, 1, 1))
        print("%-7s %-12s fwd+stats %6.1f  fwd %6.1f  dgrad %6.1f TFLOP/s" % (mode, name, fl / t1 / 1e9, fl / t0 / 1e9, fl / t2 / 1e9), flush=True)
