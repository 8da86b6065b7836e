import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args
dev = torch.device("cuda:0")
ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
torch.cuda.synchronize()
out = []
for i in range(40):
    t0 = time.perf_counter()
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
    if (i + 1) % 16 == 0:
        opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % v for v in out))
