import sys, os, time, copy, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy
from litemkd_amd.model.backbone import resnet as R
dev = torch.device("cuda", 0)
cfg = default_args(device=dev, training_iterations=10**9, print_freq=10**9)
torch.manual_seed(0)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
dist = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=1, device=dev)
pool = [src.episode(e) for e in range(2)]
trunk_copy = copy.deepcopy(student.backbone.resnet)
def two_calls_shadow(trunk, head, context_frames, target_frames):
    main = torch.cuda.current_stream(); side = ops.side_stream(context_frames.device)
    side.wait_stream(main)
    def layers_of(tk):
        ls = [lambda t, tk=tk: ops.StemFn.apply(t, getattr(tk, "0").weight, *getattr(tk, "1").args(), tk.training)]
        for name, _, _, _ in R.STAGES: ls += list(getattr(tk, name))
        return ls + [head]
    la, lb = layers_of(trunk), layers_of(trunk_copy)
    cf, tf = context_frames, target_frames
    q_upd, s_upd = [], []
    for fa, fb in zip(la, lb):
        ops.set_defer(s_upd); cf = fa(cf)
        with torch.cuda.stream(side):
            ops.set_defer(q_upd); tf = fb(tf)
    ops.set_defer(None)
    main.wait_stream(side)
    return cf, tf
def run(n):
    for i in range(n): TL.train_task(pool[i % 2], student, teacher, dist, aggregate_accuracy, cfg)
    torch.cuda.synchronize()
M2 = sys.modules["litemkd_amd.model.backbone.resnet18_2fc"]
orig = M2.two_trunk_calls
for name, f in (("shared-params", orig), ("shadow-copy", two_calls_shadow), ("shared-params", orig), ("shadow-copy", two_calls_shadow)):
    M2.two_trunk_calls = f
    run(3)
    t0 = time.perf_counter(); run(8); dt = time.perf_counter() - t0
    print(name, "%.2f ms/episode" % (dt / 8 * 1e3), flush=True)
