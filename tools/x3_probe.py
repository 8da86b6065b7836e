"""One conv layer shape, one arithmetic mode, a few launches: the target of `rocprofv3 --pmc ... -- python tools/x3_probe.py`.
usage: x3_probe.py MODE TILE Cin H Cout K stride pad [dgrad]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd
from litemkd_amd import ops
from litemkd_amd._lib import lib
mode, tile, Cin, H, Cout, K, s, p = sys.argv[1], int(sys.argv[2]), *[int(v) for v in sys.argv[3:9]]
dgrad = len(sys.argv) > 9
dev = torch.device("cuda", 0)
ops.set_conv_compute_dtype(mode)
lib().call("lmkd_conv_set_tile", tile)
Ho = (H + 2 * p - K) // s + 1
x = torch.relu(torch.randn(200, H, H, Cin, device=dev)); gy = torch.randn(200, Ho, Ho, Cout, device=dev)
w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
wp = ops._pack_weights(w, Cin, 1 if dgrad else 0)
for _ in range(4):
    if dgrad: ops.conv_bwd_data(gy, wp, (200, H, H, Cin), Cout, K, K, s, p)
    else: ops.conv_fwd(x, wp, Cout, K, K, s, p, False)
torch.cuda.synchronize()
