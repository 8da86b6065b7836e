import sys, os, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
N = 200
for (Cin, H, Cout, K, s, p) in [(128,28,128,3,1,1),(64,56,64,3,1,1),(512,7,512,3,1,1)]:
    x = torch.relu(torch.randn(N, H, H, Cin, device=dev)); w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
    gy = torch.randn(N, H, H, Cout, device=dev)
    wp = ops.pack_weights(w, Cin, 0)
    for _ in range(6):
        ops.conv_fwd(x, wp, Cout, K, K, s, p, True)
        ops.conv_bwd_weight(x, gy, tuple(w.shape), s, p)
    torch.cuda.synchronize()
