"""Per-layer timing of the conv kernels (forward / data-grad per tile config, weight-grad) on the 200-frame ResNet-18
shapes: `gpurun -- python tools/conv_bench.py`.  Tuning aid, not part of the product path."""
import sys, os, time, torch
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
N = 200
for (Cin, H, Cout, K, s, p) in [(64,56,64,3,1,1),(64,56,128,3,2,1),(128,28,128,3,1,1),(128,28,256,3,2,1),(256,14,256,3,1,1),(256,14,512,3,2,1),(512,7,512,3,1,1)]:
    x = torch.relu(torch.randn(N, H, H, Cin, device=dev)); w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
    Ho = (H + 2 * p - K) // s + 1
    gy = torch.randn(N, Ho, Ho, Cout, device=dev)
    wp, wd = ops.pack_weights(w, Cin, 0), ops.pack_weights(w, Cin, 1)
    fl = 2.0 * N * Ho * Ho * Cout * Cin * K * K
    line = "conv Cin%3d H%2d Cout%3d k%d s%d:" % (Cin, H, Cout, K, s)
    for cfg in (0, 3, 5, 6, 0):
        lib().call("lmkd_conv_set_tile", cfg)
        t1 = tm(lambda: ops.conv_fwd(x, wp, Cout, K, K, s, p, True))
        t2 = tm(lambda: ops.conv_bwd_data(gy, wd, (N, H, H, Cin), Cout, K, K, s, p))
        line += "  [cfg%d fwd %5.1f dgrad %5.1f]" % (cfg, fl/t1/1e9, fl/t2/1e9)
    lib().call("lmkd_conv_set_tile", 0)
    t3 = tm(lambda: ops.conv_bwd_weight(x, gy, tuple(w.shape), s, p))
    print(line + "  wgrad %5.1f TF" % (fl/t3/1e9), flush=True)
