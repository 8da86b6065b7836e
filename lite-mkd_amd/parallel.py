"""Episode-parallel data parallelism.  The reference's only multi-GPU mode is nn.DataParallel around
backbone.resnet (model_select.py:205-207), which splits one episode's frames over GPUs and so changes
the BatchNorm batches.  Episodes are independent (trainwandb.py:122-143), so here every GPU (one
process each) runs whole episodes and the ranks exchange exactly one thing: the flat fp32 gradient
bucket, summed with ONE RCCL all-reduce over xGMI per optimizer step (`torch.distributed` backend
"nccl" is RCCL on ROCm).  BatchNorm statistics stay per-episode-local as in the 1-GPU reference."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun).  -> (rank, world, device)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    backend = backend or os.environ.get("LMKD_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    if use_cuda:
        ndev = torch.cuda.device_count()
        if local >= ndev:
            if backend != "gloo":
                # RCCL needs one device per rank: two ranks of one communicator on the same GPU hang or fail deep inside RCCL
                raise RuntimeError("LOCAL_RANK %d but only %d visible GPU(s): the %s (RCCL) backend needs one GPU per rank; "
                                   "several ranks may share a card only with LMKD_DIST_BACKEND=gloo (rehearsals)" % (local, ndev, backend))
            local = local % ndev                       # gloo rehearsal: several ranks on one card
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


ALLREDUCE_TIMING = None


def sync_bn_running_stats(module):
    """COLLECTIVE (every rank must call it): {state_dict key: tensor} of every BatchNorm running_mean / running_var of `module`
    pooled over the ranks (each rank's BatchNorm sees only its own episodes; the estimates are pooled when a
    checkpoint is written).  running_mean = mean over the ranks; running_var = the variance of the pooled population,
    mean_r(var_r) + mean_r(mean_r^2) - (mean_r mean_r)^2, i.e. the ranks' own variances plus the spread of their means.
    ONE all-reduce of the concatenated buffers on the current stream.  The live buffers are not modified."""
    sd = module.state_dict()
    mkeys = [k for k in sd if k.endswith("running_mean")]
    if not mkeys:
        return {}
    vkeys = [k[:-len("running_mean")] + "running_var" for k in mkeys]
    means = torch.cat([sd[k].detach().reshape(-1).float() for k in mkeys])
    varis = torch.cat([sd[k].detach().reshape(-1).float() for k in vkeys])
    flat = torch.cat([means, varis, means * means])
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world_size()
    n = means.numel()
    m, v, m2 = flat[:n], flat[n:2 * n], flat[2 * n:]
    v = v + (m2 - m * m).clamp_min(0.0)
    out, o = {}, 0
    for km, kv in zip(mkeys, vkeys):
        c = sd[km].numel()
        out[km] = m[o:o + c].reshape(sd[km].shape).clone()
        out[kv] = v[o:o + c].reshape(sd[kv].shape).clone()
        o += c
    return out


class FlatParams:
    """All trainable parameters of a module as views into ONE contiguous fp32 buffer, and their .grad as
    views into ONE gradient buffer (22.7 M floats = 90.9 MB for the default student): the all-reduce
    bucket and the fused optimizer both work on these two flat buffers.  Each parameter starts on a
    16-byte boundary (float4 loads in the GEMM / conv loaders)."""

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, self.offsets = params, offs
        # second gradient buffer for kernels that accumulate parameter gradients directly from the side stream (ops.DIRECT_PARAM_GRAD:
        # the query-frame trunk call's BatchNorm backward); folded into `grad` before every use of it (fold_shadow)
        self.shadow = torch.zeros(total, dtype=torch.float32, device=dev) if dev.type == "cuda" else None
        self.shadow_dirty = False
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = self.flat[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                p.grad = self.grad[o:o + p.numel()].view_as(p)
                if self.shadow is not None:
                    from . import ops
                    ops.register_grad_slot(p, self.shadow[o:o + p.numel()].view_as(p))
        self.numel = total

    def zero_grad(self):
        self.grad.zero_()
        if self.shadow is not None:
            self.shadow.zero_()

    def fold_shadow(self):
        """grad += shadow; shadow = 0 (call on the stream that owns the gradient buffer, after the side streams have been joined)"""
        if self.shadow is not None:
            self.grad.add_(self.shadow)
            self.shadow.zero_()

    def allreduce_grads(self, upto=None):
        """One all-reduce(sum) of the whole bucket (or of its first `upto` elements: the tail went out early, EarlyAllReduce).  No-op on a
        single process.  With ALLREDUCE_TIMING set to a list, the collective is bracketed by two events on the current stream (bench.py
        reports the mean)."""
        if world_size() > 1:
            g = self.grad if upto is None or upto >= self.numel else self.grad[:upto]
            if g.numel() == 0:
                return
            if ALLREDUCE_TIMING is not None and self.grad.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                e1.record()
                ALLREDUCE_TIMING.append((e0, e1, "at step"))
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)

    def broadcast_params(self, src=0):
        if world_size() > 1:
            dist.broadcast(self.flat, src=src)


class EarlyAllReduce:
    """Overlap the gradient all-reduce with the backward pass of the last episode before an optimizer step (trainwandb.py:141-143).

    The backward pass reaches the parameters in reverse order: matcher / fc heads first, then the trunk from its last stage down to the
    stem.  In the flat bucket (module.parameters() order) the last trunk stage and everything after it - 19.9 of the 22.7 M gradient
    elements of the default student - are ONE contiguous tail [split, numel), and that tail is final once the backward has left the last
    stage, while 3/4 of the trunk's backward time (stages 1-3, the stem) is still ahead.  A tensor hook on the last stage's input (set in
    the trunk's forward when `armed`) fires at that moment - once per trunk call - and the all-reduce of the tail is issued from a
    communication stream that waits for every stream that may have written it (weight-gradient, side, auxiliary streams) and folds the
    tail of the shadow buffer first.  FusedOptimizer.step() then waits for it and all-reduces only the head [0, split) itself.
    On an 8-GPU ring over xGMI the 80 MB tail is ~1.5 - 3 ms of a 55 ms optimizer interval (world 8: a step every 2 local episodes);
    the 11 MB head ~0.3 ms."""

    def __init__(self, bucket, module, tail_prefix=("backbone.resnet.7.",)):
        self.bucket = bucket
        names = [n for n, p in module.named_parameters() if p.requires_grad]
        assert len(names) == len(bucket.params)
        first = next((i for i, n in enumerate(names) if n.startswith(tuple(tail_prefix))), None)
        self.split = bucket.offsets[first] if first is not None else bucket.numel
        self.armed = False
        self.count = self.expected = 0
        self.work = None
        self.stream = None
        self.events = None

    def arm(self):
        """the next backward pass completes the gradients of an optimizer interval.  The hook firings to wait for are COUNTED while the
        armed forward registers them (registered(): two trunk calls of an episode, one when they run merged - which the trunk decides
        per call, so a global flag cannot tell)"""
        self.armed = world_size() > 1 and self.split < self.bucket.numel and self.bucket.grad.is_cuda
        self.count, self.expected, self.work = 0, 0, None

    def registered(self):
        """the trunk's forward has put `hook` on one more tensor"""
        self.expected += 1

    def hook(self, grad):
        """tensor hook on the last stage's input (returns None: the gradient passes unchanged)"""
        if not self.armed:
            return None
        self.count += 1
        if self.count >= self.expected:      # every hook the armed forward registered has fired: the tail is final
            self.launch()
        return None

    def launch(self):
        from . import ops
        self.armed = False
        b = self.bucket
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=b.grad.device)
        cs = self.stream
        cs.wait_stream(torch.cuda.current_stream())          # the stream of the node that just ran (the trunk call's own)
        for table in (ops._WG_STREAM, ops._side_streams, ops._aux_streams, ops._lane_mains):
            for s in table.values():
                cs.wait_stream(s)
        cs.wait_stream(torch.cuda.default_stream(b.grad.device))
        with torch.cuda.stream(cs):
            tail = b.grad[self.split:]
            if b.shadow is not None:
                sh = b.shadow[self.split:]
                tail.add_(sh)
                sh.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cs)
            self.work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
            self.work.wait()                                   # (orders the communication stream behind the collective; the host does not block)
            e1.record(cs)
            self.events = (e0, e1)

    def finish(self):
        """-> number of leading bucket elements the caller still has to all-reduce (the whole bucket if nothing was sent early)"""
        self.armed = False
        if self.work is None:
            return self.bucket.numel
        torch.cuda.current_stream().wait_stream(self.stream)
        self.work = None
        if ALLREDUCE_TIMING is not None and self.events is not None:
            ALLREDUCE_TIMING.append(self.events + ("early",))
        return self.split
