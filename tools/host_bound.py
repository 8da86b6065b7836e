"""Is the episode loop host-bound?  Queues N training episodes without any fence and reports (a) the host time to enqueue them,
(b) the wall time until the GPU has drained them.  If (a) ~ (b) the Python/launch side is the limit, if (a) << (b) the GPU is.
usage: python tools/host_bound.py [f32|f32native|bf16] [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
ops.set_conv_compute_dtype({"f32": "fp32x3", "f32native": "fp32", "bf16": "bf16"}[mode])
ops.set_activation_dtype("bf16" if mode == "bf16" else "fp32")
ops.SIDE_WGRAD = True
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
opt = TL.FusedOptimizer(student, cfg.opt, cfg.learning_rate)
distiller = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=2024, rank=0, device=dev)
pool = [src.episode(e) for e in range(4)]


def run(k):
    for i in range(k):
        TL.train_task(pool[i % 4], student, teacher, distiller, aggregate_accuracy, cfg)
        if (i + 1) % cfg.tasks_per_batch == 0:
            opt.step(); opt.zero_grad()


run(4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("mode %s: host enqueue %.2f ms/episode, wall %.2f ms/episode (%d episodes)" % (mode, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n))
# host cost alone: the same loop with the GPU kept idle between episodes
hs = 0.0
for i in range(8):
    torch.cuda.synchronize(); a = time.perf_counter()
    TL.train_task(pool[i % 4], student, teacher, distiller, aggregate_accuracy, cfg)
    hs += time.perf_counter() - a
torch.cuda.synchronize()
print("host time of one episode's enqueue with an empty queue: %.2f ms" % (hs / 8 * 1e3))
