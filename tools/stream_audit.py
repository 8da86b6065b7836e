"""Stream-lifetime audit of the benchmark's schedule (lite-mkd_amd/_audit.py): a few 400-frame training episodes with an optimizer step under
Schedule.bench() (merged trunk call, cross-episode pipelining, side weight-gradient stream) and under Schedule.two_call(), in fp32h2, with
every entry-point call checked: a tensor used on a stream other than the one it was allocated on must carry a record_stream for it.
usage: LMKD_STREAM_AUDIT=1 python tools/stream_audit.py [img]   -> findings grouped by (entry point, argument)"""
import os, sys
os.environ["LMKD_STREAM_AUDIT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import litemkd_amd  # noqa: F401
from litemkd_amd import ops, trainloop as TL, _audit
from litemkd_amd.options import default_args
from litemkd_amd.schedule import Schedule

img = int(sys.argv[1]) if len(sys.argv) > 1 else 224
if "--selftest" in sys.argv:
    # negative control: the state of round 4's first fp32h2 version - operands recorded on the weight-gradient stream, the words of their
    # maxima (int32) not: the audit must name lmkd_conv2d_bwd_weight_seg
    _rec = torch.Tensor.record_stream
    torch.Tensor.record_stream = lambda self, s: None if self.dtype is torch.int32 else _rec(self, s)
dev = torch.device("cuda", 0)
total = 0
for sched in (Schedule.bench(conv_dtype="fp32h2"), Schedule.two_call(conv_dtype="fp32h2")):
    cfg = default_args(shot=5, device=dev, trans_dropout=0.1, training_iterations=7, print_freq=3, tasks_per_batch=4, img_size=img)
    torch.manual_seed(0)
    with sched.applied():
        student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg, base_seed=11)
        _audit.arm()      # the model, the optimizer and the flat buffers exist: everything allocated from here on is judged
        TL.train(student, teacher, src, distiller, opt, sch, acc_fn, cfg, log=lambda *a: None, schedule=sched)
    torch.cuda.synchronize()
    f = _audit.summary()
    print("schedule", "bench" if sched.pipeline_episodes else "two_call", ":", len(_audit.findings()), "findings")
    for cnt, name, arg, shape in f:
        print("  %5d x %s argument %d e.g. %s" % (cnt, name, arg, shape))
    total += len(_audit.findings(clear=True))
    _audit.arm(False)
print("total findings", total)
