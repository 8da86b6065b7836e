"""soak: many episodes with optimizer steps; allocator statistics must stay flat (no leak), loss finite.  usage: python tools/soak.py [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
from litemkd_amd.schedule import Schedule
getattr(Schedule, os.environ.get("LMKD_SCHED", "bench"))(conv_dtype=os.environ.get("LMKD_CONV", "fp32h2")).apply()      # bench.py's schedule and arithmetic by default (the loop below is the unpipelined one); LMKD_SCHED = two_call | serial
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
t0 = time.perf_counter()
for i in range(n):
    loss, acc, _ = TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
    if (i + 1) % 16 == 0:
        opt.step()
        opt.zero_grad()
    sch.step()
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats()
        print("episode %4d  loss %.4f  %.1f episodes/s  reserved %.2f GB  allocated %.2f GB  device mallocs %d" % (
            i + 1, float(loss), 100 / (time.perf_counter() - t0), st["reserved_bytes.all.current"] / 2 ** 30, st["allocated_bytes.all.current"] / 2 ** 30,
            st["num_device_alloc"]), flush=True)
        t0 = time.perf_counter()
