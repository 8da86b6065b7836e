"""cProfile of the host side of the eager episode loop (where do the ~10 ms of Python per episode go?)
usage: python tools/host_profile.py [episodes]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
for i in range(4):
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print("per episode: %.2f ms of profiled host time" % (st.total_tt / n * 1e3))
st.print_stats(28)
