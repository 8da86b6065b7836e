"""Which launches of a training episode are NOT liblmkd_hip.so kernels, and where in the Python code do they come from?
Runs a few benchmark episodes under torch.profiler with stacks and prints, per (ATen op, innermost repo frame), the number of
device launches per episode.  usage: python tools/aten_ops.py [f32|f32x3|bf16] [episodes]   (f32 = bench.py's headline arithmetic, fp32h2)"""
import os
import sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
ops.set_conv_compute_dtype({"f32": "fp32h2", "f32x3": "fp32x3", "bf16": "bf16"}[mode])
ops.set_activation_dtype("bf16" if mode == "bf16" else "fp32")
from litemkd_amd.schedule import Schedule
Schedule.from_env().apply(arithmetic=False)      # the benchmark's schedule (merged trunk call; the episodes here are not pipelined: one at a time)
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
opt = TL.FusedOptimizer(student, cfg.opt, cfg.learning_rate)
distiller = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=2024, rank=0, device=dev)
pool = [src.episode(e) for e in range(2)]
for i in range(3):
    TL.train_task(pool[i % 2], student, teacher, distiller, aggregate_accuracy, cfg)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(n):
        TL.train_task(pool[i % 2], student, teacher, distiller, aggregate_accuracy, cfg)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ops_c, kern_c = Counter(), Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern_c[ev.name[:90]] += 1
        continue
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    if any(k for k in ev.kernels):
        st = [f.strip() for f in (ev.stack or [])]
        fr = [f for f in st if "lite-mkd_amd" in f or "litemkd" in f or "bench.py" in f or "tools/" in f]
        ops_c[(ev.name, (fr[0] if fr else " <- ".join(st[:3]) or "(no python stack: autograd engine)")[-150:])] += len(ev.kernels)
print("device launches per episode by kernel name (top 60):")
for k, v in kern_c.most_common(60):
    print("  %6.1f  %s" % (v / n, k))
print("total launches per episode: %.1f" % (sum(kern_c.values()) / n))
print("ATen ops that launch device work, per episode, by innermost repo frame:")
for (name, fr), v in ops_c.most_common(80):
    print("  %6.1f  %-28s %s" % (v / n, name, fr))
