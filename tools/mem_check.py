"""Long-run check: memory stays flat and throughput steady over many training episodes (three-stream schedule, caching
allocator + record_stream).  `gpurun -- python tools/mem_check.py [episodes] [fp32x3|bf16]`"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import litemkd_amd
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.model.model_select import Student, Teacher
from litemkd_amd.distillers import Distiller
from litemkd_amd.options import default_args
from litemkd_amd.utils import aggregate_accuracy
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
if len(sys.argv) > 2:      # arithmetic mode of the convolutions (default fp32 MFMA); "fp32x3" = the benchmark's headline
    ops.set_conv_compute_dtype(sys.argv[2])
ops.SIDE_WGRAD = True
cfg = default_args(shot=5, device=dev, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(0)
student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
opt = TL.FusedOptimizer(student, cfg.opt, cfg.learning_rate)
dist = Distiller(cfg.distill_name, cfg.cfg, dev)
src = TL.SyntheticEpisodes(cfg, base_seed=1, device=dev)
pool = [src.episode(e) for e in range(2)]
ops.SYNC_WGRAD_AT_BACKWARD_END = False
t0 = time.perf_counter()
for it in range(1, n + 1):
    loss, acc, _ = TL.train_task(pool[it % 2], student, teacher, dist, aggregate_accuracy, cfg)
    if (it + 1) % 16 == 0:
        opt.step(); opt.zero_grad()
    if it % 16 == 0:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        print("episode %3d  %.1f episodes/s  loss %.4f  allocated %.2f GB  reserved %.2f GB  peak %.2f GB" % (
            it, 16 / dt, float(loss), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            torch.cuda.max_memory_allocated() / 2**30), flush=True)
