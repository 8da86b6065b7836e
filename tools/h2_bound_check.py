"""how loose is the bound of the loader-side BatchNorm (lmkd_bn_finalize_bound) on the trunk?  One 400-frame training forward in fp32h2:
per block, bound / true maximum of relu(bn1(c1)) per frame segment."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args
from litemkd_amd.schedule import Schedule
dev = torch.device("cuda:0")
Schedule.bench(conv_dtype="fp32h2").apply()
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
ep = src.episode(0)
ops.BLOCK_TAPS = []
TL.train_task(ep, student, teacher, distiller, acc_fn, cfg)
taps, ops.BLOCK_TAPS = ops.BLOCK_TAPS, None
torch.cuda.synchronize()
k = 0
for t in taps:
    if "c1" not in t:
        continue
    k += 1
    c1, st1, seg = t["c1"], t["st1"], t["seg"]
    w = getattr(c1, "_lmkd_pre_amax", None)
    if w is None:
        print("block %d: no bound recorded" % k)
        continue
    a1 = ops.bn_apply(c1, st1, True, seg=seg)
    f = w.view(torch.float32)
    half = f.numel() // 2
    out = []
    for i, p in enumerate([a1[:seg], a1[seg:]] if seg else [a1]):
        b, tr = float(f[i * half:(i + 1) * half].max()), float(p.max())
        out.append("bound %.3g / true %.3g = 2^%.2f" % (b, tr, math.log2(b / tr)))
    print("block %d  %s   %s" % (k, tuple(c1.shape), "   |   ".join(out)))
