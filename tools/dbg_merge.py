"""debug aid: merged trunk call vs two calls, tap by tap (first tensor that differs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops
from litemkd_amd.model.backbone import resnet as R
dev = torch.device("cuda", 0)
Fs, Fq, img = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(5)
trunk = R.ResNet18Trunk().to(dev).train()
g = torch.Generator().manual_seed(6)
cf, tf = torch.rand(Fs, 3, img, img, generator=g).to(dev), torch.rand(Fq, 3, img, img, generator=g).to(dev)
st0 = {k: v.clone() for k, v in trunk.state_dict().items()}
taps = {}
for merged in (False, True):
    trunk.load_state_dict(st0)
    R.MERGE_TRUNK_CALLS, R.OVERLAP_TRUNK_CALLS = merged, False
    ops.BLOCK_TAPS = []
    with torch.no_grad():
        X, _ = R.trunk_features(trunk, ops.PoolHeadFn.apply, cf, tf)
    taps[merged] = ops.split_block_taps(ops.BLOCK_TAPS)
    ops.BLOCK_TAPS = None
    torch.cuda.synchronize()
print(len(taps[False]), len(taps[True]))
for i, (a, b) in enumerate(zip(taps[False], taps[True])):
    for k in a:
        eq = torch.equal(a[k], b[k])
        if not eq:
            d = (a[k].float() - b[k].float()).abs()
            print("tap", i, k, tuple(a[k].shape), "max diff", float(d.max()), "n diff", int((d > 0).sum()))
            if a[k].dim() == 2:
                print("  rows differing", (d > 0).any(1).nonzero().flatten().tolist())
