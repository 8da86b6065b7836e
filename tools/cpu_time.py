"""Host-side cost of one episode: wall vs process CPU time (main + autograd thread) in the overlapped and serial schedules."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy
from litemkd_amd.model.backbone import resnet as R
dev = torch.device("cuda", 0)
cfg = default_args(device=dev, training_iterations=10**9, print_freq=10**9, img_size=int(os.environ.get("IMG", "224")))
torch.manual_seed(0)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
opt = TL.FusedOptimizer(student, "sgd", 1e-4)
dist = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=1, device=dev)
pool = [src.episode(e) for e in range(2)]
for mode in (True, False, True):
    R.OVERLAP_TRUNK_CALLS = mode
    for i in range(3): TL.train_task(pool[i % 2], student, teacher, dist, aggregate_accuracy, cfg)
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    for i in range(8): TL.train_task(pool[i % 2], student, teacher, dist, aggregate_accuracy, cfg)
    t1, c1 = time.perf_counter(), time.process_time()
    torch.cuda.synchronize()
    t2, c2 = time.perf_counter(), time.process_time()
    print("overlap", mode, "wall %.1f ms/ep | enqueue-wall %.1f | process CPU %.1f ms/ep (during enqueue %.1f)" % ((t2 - t0) / 8e-3, (t1 - t0) / 8e-3, (c2 - c0) / 8e-3, (c1 - c0) / 8e-3), flush=True)
