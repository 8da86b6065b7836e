import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
import sys as _s; _m = _s.modules["litemkd_amd.model.backbone.resnet18_2fc"]
class M: pass
M.resnet18_2fc = _m.resnet18_2fc
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy
dev = torch.device("cuda", 0)
cfg = default_args(device=dev, training_iterations=10**9, print_freq=10**9)
torch.manual_seed(0)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
opt = TL.FusedOptimizer(student, "sgd", 1e-4)
dist = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=1, device=dev)
pool = [src.episode(e) for e in range(2)]
side = torch.cuda.Stream()
orig_forward = M.resnet18_2fc.forward
def fwd2(self, context_feature, context_labels, target_feature):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        tf = ops.PoolHeadFn.apply(self.resnet(target_feature))
    cf = ops.PoolHeadFn.apply(self.resnet(context_feature))
    main.wait_stream(side)
    L, D = self.args.seq_len, 2048
    return ({"context_features_1": self.fc1(cf).reshape(-1, L, D), "context_features_2": self.fc2(cf).reshape(-1, L, D)},
            {"target_features_1": self.fc1(tf).reshape(-1, L, D), "target_features_2": self.fc2(tf).reshape(-1, L, D)})
def run(n):
    for i in range(n):
        TL.train_task(pool[i % 2], student, teacher, dist, aggregate_accuracy, cfg)
    torch.cuda.synchronize()
for name, f in (("serial", orig_forward), ("two-stream", fwd2), ("serial", orig_forward), ("two-stream", fwd2)):
    M.resnet18_2fc.forward = f
    run(2)
    t0 = time.perf_counter(); run(8); dt = time.perf_counter() - t0
    print(name, "%.2f ms/episode" % (dt / 8 * 1e3), flush=True)
