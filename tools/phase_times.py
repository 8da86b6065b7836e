"""GPU time of the phases of an (un-profiled, overlapped) training episode, from HIP events on the episode's main stream:
trunk forward | heads + loss forward | heads backward (until the gradient reaches the trunk's features) | trunk backward on the main
stream | tail: what the weight-gradient stream still runs after the main stream's last kernel.
usage: python tools/phase_times.py [episodes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL
from litemkd_amd.options import default_args

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
ops.set_conv_compute_dtype(os.environ.get("LMKD_CONV", "fp32h2"))      # bench.py's headline arithmetic
cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=10 ** 9, print_freq=10 ** 9)
torch.manual_seed(1234)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=2024)
pool = [src.episode(e) for e in range(2)]
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev[name] = e


def cls_pre(mod, inp):
    mark("trunk_fwd_end")
    for a in inp:
        for t in (a.values() if isinstance(a, dict) else [a]):
            if torch.is_tensor(t) and t.is_floating_point() and t.requires_grad:
                t.register_hook(lambda g, _t=t: (mark("heads_bwd_end_%d" % id(_t)), g)[1])


student.classifier.register_forward_pre_hook(cls_pre)
for i in range(4):
    TL.train_task(pool[i % 2], student, teacher, distiller, acc_fn, cfg)
torch.cuda.synchronize()
rows = []
for i in range(n):
    ev.clear()
    mark("start")
    prepared = TL.prepare_task(pool[i % 2], cfg.device)
    loss, acc = TL._episode_forward(prepared, student, teacher, distiller, acc_fn, cfg)
    mark("fwd_end")
    loss.backward()
    mark("bwd_main_end")
    ops.wait_weight_grads()
    mark("all_end")
    torch.cuda.synchronize()
    hb = [e for k, e in ev.items() if k.startswith("heads_bwd_end")]
    t = lambda a, b: a.elapsed_time(b)
    hb_last = max(hb, key=lambda e: ev["start"].elapsed_time(e))
    rows.append((t(ev["start"], ev["trunk_fwd_end"]), t(ev["trunk_fwd_end"], ev["fwd_end"]), t(ev["fwd_end"], hb_last),
                 t(hb_last, ev["bwd_main_end"]), t(ev["bwd_main_end"], ev["all_end"]), t(ev["start"], ev["all_end"])))
rows = rows[2:]
avg = [sum(r[j] for r in rows) / len(rows) for j in range(6)]
print("ms per episode (episodes run one at a time: the queue drains between them): trunk forward %.2f | heads + loss forward %.2f | heads backward %.2f | "
      "trunk backward (main stream) %.2f | weight-gradient stream tail %.2f | total %.2f" % tuple(avg))
