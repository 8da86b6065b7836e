"""One-rank rehearsal of every RCCL call the multi-GPU path makes (the box has one GPU; two ranks of one communicator cannot share it):
process group "nccl" with world_size 1, then - on the real student's flat gradient bucket - parallel.EarlyAllReduce.launch() / finish()
(async all-reduce of the bucket's tail from a communication stream), FlatParams.allreduce_grads(head), broadcast_params, the float64 MAX
reductions, all_gather_object, barrier and sync_bn_running_stats as bench.py / trainloop issue them.  Checks API validity and stream
semantics (values unchanged by a one-rank sum), not bandwidth.  usage: python tools/rccl_rehearsal.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1)
import litemkd_amd  # noqa: F401
from litemkd_amd import parallel as PAR, trainloop as TL
from litemkd_amd.options import default_args
PAR.world_size = lambda: 2          # take the world > 1 branches (the communicator itself has one rank: sums leave the values unchanged)
cfg = default_args(shot=1, query_per_class=1, img_size=64, device=dev)
torch.manual_seed(0)
student, teacher, src, distiller, acc_fn, _, opt, sch = TL.make(cfg)
b = opt.bucket
b.grad.normal_()
b.shadow.normal_()
ref = (b.grad + b.shadow).clone()
PAR.ALLREDUCE_TIMING = []
opt.early.arm()
opt.early.registered()          # (the trunk's forward does this for every tensor it puts the hook on)
assert opt.early.armed
opt.early.hook(None)                # fires launch(): tail += shadow tail, async all-reduce on the communication stream
assert opt.early.work is not None
upto = opt.early.finish()
b.grad[:upto].add_(b.shadow[:upto])
b.shadow[:upto].zero_()
b.allreduce_grads(upto)
torch.cuda.synchronize()
assert torch.equal(b.grad, ref), float((b.grad - ref).abs().max())
assert float(b.shadow.abs().max()) == 0.0
ms = [(e[0].elapsed_time(e[1]), e[2]) for e in PAR.ALLREDUCE_TIMING]
b.broadcast_params(0)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
names = [None]
dist.all_gather_object(names, "rank 0: %s" % torch.cuda.get_device_name(dev))
dist.barrier()
sd = PAR.sync_bn_running_stats(student)
torch.cuda.synchronize()
print("RCCL one-rank rehearsal ok: backend %s, early all-reduce of %d of %d elements, timings %s, gathered %s, %d pooled BatchNorm buffers" % (
    dist.get_backend(), b.numel - opt.early.split, b.numel, [("%.3f ms" % m, k) for m, k in ms], names, len(sd)))
dist.destroy_process_group()
