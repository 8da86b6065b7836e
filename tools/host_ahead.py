"""Is the host AHEAD of the GPU at episode boundaries?  At the top of every episode: has the GPU already finished the previous
episode (event.query())?  If yes the queue ran dry - something on the host side waits for the GPU once per episode.  Also prints
the caching allocator's device-malloc count (a hipMalloc / hipFree in steady state synchronises) and host time per phase.
usage: python tools/host_ahead.py [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
ops.set_conv_compute_dtype(os.environ.get("LMKD_CONV", "fp32h2"))      # bench.py's headline arithmetic
TL.TEACHER_STREAM = os.environ.get("LMKD_TEACHER_STREAM", "1") != "0"
ops.HEADS_ON_TWO_STREAMS = os.environ.get("LMKD_HEADS2", "1") != "0"
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
for i in range(4):
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
prev_ev, t_prev = None, time.perf_counter()
dry = 0
for i in range(n):
    done = prev_ev.query() if prev_ev is not None else None
    dry += 1 if done else 0
    t0 = time.perf_counter()
    prepared = TL.prepare_task(pool[i % 2], cfg.device)
    t1 = time.perf_counter()
    loss, acc = TL._episode_forward(prepared, student, teacher, distiller, acc_fn, cfg)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    prev_ev = torch.cuda.Event()
    prev_ev.record()
    print("episode %2d: previous episode already finished on the GPU at its start: %-5s | host ms: prepare %.2f forward %.2f backward %.2f | since last start %.2f"
          % (i, done, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t0 - t_prev) * 1e3), flush=True)
    t_prev = t0
torch.cuda.synchronize()
st1 = torch.cuda.memory_stats()
print("queue ran dry at %d of %d episode starts; device mallocs during the loop: %d, frees: %d, alloc retries: %d" % (
    dry, n - 1, st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"], st1["num_alloc_retries"] - st0["num_alloc_retries"]))
