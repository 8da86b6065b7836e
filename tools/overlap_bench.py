"""Co-execution experiment: 40 convolutions and 40 BatchNorm-backward launches alone, and on two streams at once
(TILE=<id> selects the conv tile).  Result recorded in DESIGN.md section 5."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops
dev = torch.device("cuda", 0)
N, Cin, H, Cout = 200, 128, 28, 128
x = torch.relu(torch.randn(N, H, H, Cin, device=dev)); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
wp = ops._pack_weights(w, Cin, 0)
big = torch.randn(200, 56, 56, 64, device=dev); big2 = torch.relu(torch.randn_like(big)); gy = torch.randn_like(big)
C = 64
part = torch.stack([big.reshape(-1, C).sum(0, keepdim=True), (big.reshape(-1, C) ** 2).sum(0, keepdim=True)], -1).contiguous()
st = ops.bn_stats_train(part, big.numel() // C, torch.ones(C, device=dev), torch.zeros(C, device=dev), None, None)
gam = torch.ones(C, device=dev)
def convs(n):
    for _ in range(n): ops.conv_fwd(x, wp, Cout, 3, 3, 1, 1, True)
def bns(n):
    for _ in range(n): ops.bn_backward(gy, big, big2, st, gam, 1, want_g=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fa, fb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): fa()
    if fb is not None:
        with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
from litemkd_amd._lib import lib
lib().call("lmkd_conv_set_tile", int(os.environ.get("TILE", "0")))
lib().call("lmkd_set_elementwise_wg_per_cu", int(os.environ.get("EW", "4")))
convs(3); bns(3); torch.cuda.synchronize()
for _ in range(2):
    a = timed(lambda: convs(40), None); b = timed(lambda: bns(40), None)
    ab = timed(lambda: convs(40), lambda: bns(40)); aa = timed(lambda: convs(20), lambda: convs(20))
    print("conv x40 alone %.2f ms | bn_bwd x40 alone %.2f ms | both concurrently %.2f ms (sum %.2f) | conv 20+20 on two streams %.2f ms" % (a, b, ab, a + b, aa), flush=True)
