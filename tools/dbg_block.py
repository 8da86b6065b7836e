"""debug: per-tensor fp64-anchored errors of one BasicBlock at a given size"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import litemkd_amd
from litemkd_amd import ops
from litemkd_amd.model.backbone import resnet as R
from _anchor import rel_l2
dev = torch.device("cuda", 0)
N, cin, cout, H, stride = [int(v) for v in sys.argv[1:6]]
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
def nchw(x): return x.permute(0, 3, 1, 2).contiguous()
torch.manual_seed(0)
blk = R._Block(cin, cout, stride)
x0 = torch.relu(torch.randn(N, cin, H, H))
gy = torch.randn(N, cout, H // stride, H // stride)
def ref_run(dt):
    x = x0.detach().clone().to(dt).requires_grad_()
    ref = {k: v.detach().clone().to(dt).requires_grad_() for k, v in blk.named_parameters()}
    def bn(t, pre):
        return F.batch_norm(t, torch.zeros(cout, dtype=dt), torch.ones(cout, dtype=dt), ref[pre + ".weight"], ref[pre + ".bias"], True, 0.1, 1e-5)
    c1 = F.conv2d(x, ref["conv1.weight"], None, stride, 1); c1.retain_grad()
    a1 = F.relu(bn(c1, "bn1")); a1.retain_grad()
    c2 = F.conv2d(a1, ref["conv2.weight"], None, 1, 1); c2.retain_grad()
    out = bn(c2, "bn2")
    idn = x
    if blk.downsample is not None:
        idn = bn(F.conv2d(x, ref["downsample.0.weight"], None, stride, 0), "downsample.1")
    y = F.relu(out + idn)
    y.backward(gy.to(dt))
    g = {k: v.grad for k, v in ref.items()}
    g["x"] = x.grad
    return y.detach(), g, dict(c1=c1.detach(), a1=a1.detach(), dc1=c1.grad, da1=a1.grad, dc2=c2.grad)
y32, g32, t32 = ref_run(torch.float32)
y64, g64, t64 = ref_run(torch.float64)
blk = blk.to(dev).train()
xd = nhwc(x0).to(dev).requires_grad_()
yd = blk(xd)
yd.backward(nhwc(gy).to(dev))
torch.cuda.synchronize()
print("y", rel_l2(nchw(yd), y64), rel_l2(y32, y64))
hip = {k: v.grad for k, v in blk.named_parameters()}
hip["x"] = nchw(xd.grad)
for k in g64:
    print("%-24s hip %.3e cpu %.3e" % (k, rel_l2(hip[k], g64[k]), rel_l2(g32[k], g64[k])))
# op level: wgrad of conv1 with the fp64 dc1 and x
dc1 = nhwc(t64["dc1"].float()).to(dev)
dw = ops.conv_bwd_weight(xd.detach(), dc1, (cout, cin, 3, 3), stride, 1)
print("wgrad(conv1) from exact dc1: hip", rel_l2(dw, g64["conv1.weight"]))
dw2 = ops.conv_bwd_weight(nhwc(t64["a1"].float()).to(dev), nhwc(t64["dc2"].float()).to(dev), (cout, cout, 3, 3), 1, 1)
print("wgrad(conv2) from exact a1, dc2: hip", rel_l2(dw2, g64["conv2.weight"]))
import ctypes
info = (ctypes.c_int * 5)()
litemkd_amd.lib().call("lmkd_conv2d_plan", 2, N, H, H, cin, cin, cout, 3, 3, stride, 1, info); print("wgrad plan conv1", list(info))
litemkd_amd.lib().call("lmkd_conv2d_plan", 1, N, H, H, cin, cin, cout, 3, 3, stride, 1, info); print("dgrad plan conv1", list(info))
