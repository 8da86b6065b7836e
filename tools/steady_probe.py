"""episodes/s in windows of 10 episodes over a longer run: is a slow process slow throughout or only at its start?
usage: python tools/steady_probe.py [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
out = []
t0 = time.perf_counter()
for i in range(n):
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
    if (i + 1) % 16 == 0:
        opt.step()
        opt.zero_grad()
    if (i + 1) % 10 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out.append(10 / (t1 - t0))
        t0 = t1
print(" ".join("%.1f" % v for v in out))
