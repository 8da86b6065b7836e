"""ResNet-18 trunk = torchvision resnet18().children()[:-2] (resnet18_2fc.py:30-33), rebuilt on the
HIP conv/BN kernels.  Module/parameter names reproduce nn.Sequential's state_dict keys
(`0.weight`, `1.running_mean`, `4.0.conv1.weight`, `5.0.downsample.1.bias`, ...) so checkpoints
written by the reference load unchanged (model_select.py:138-153)."""
import math

import torch
import torch.nn as nn

from ... import ops

STAGES = [("4", 64, 64, 1), ("5", 64, 128, 2), ("6", 128, 256, 2), ("7", 256, 512, 2)]


class _Conv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")   # torchvision ResNet.__init__
        self.weight = nn.Parameter(w)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def args(self):
        return self.weight, self.bias, self.running_mean, self.running_var


class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = _Conv(cin, cout, 3)
        self.bn1 = _BN(cout)
        self.conv2 = _Conv(cout, cout, 3)
        self.bn2 = _BN(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential()
            self.downsample.add_module("0", _Conv(cin, cout, 1))
            self.downsample.add_module("1", _BN(cout))

    def forward(self, x, seg=0):
        """seg = F0 > 0: x holds both trunk calls of an episode, frames [0, F0) | [F0, F) (merged_trunk_call)"""
        tr = self.training
        ops.mark_cacheable(self.conv1.weight)
        ops.mark_cacheable(self.conv2.weight)
        if self.downsample is not None:
            ops.mark_cacheable(self.downsample[0].weight)
            ds = (self.downsample[0].weight,) + self.downsample[1].args()
        else:
            ds = (None, None, None, None, None)
        return ops.BasicBlockFn.apply(x, self.stride, tr, self.conv1.weight, *self.bn1.args(),
                                      self.conv2.weight, *self.bn2.args(), *ds, seg)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        self.stride = stride
        cout = width * 4
        self.conv1 = _Conv(cin, width, 1)
        self.bn1 = _BN(width)
        self.conv2 = _Conv(width, width, 3)
        self.bn2 = _BN(width)
        self.conv3 = _Conv(width, cout, 1)
        self.bn3 = _BN(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential()
            self.downsample.add_module("0", _Conv(cin, cout, 1))
            self.downsample.add_module("1", _BN(cout))

    def forward(self, x, seg=0):
        for c in (self.conv1, self.conv2, self.conv3):
            ops.mark_cacheable(c.weight)
        if self.downsample is not None:
            ops.mark_cacheable(self.downsample[0].weight)
            ds = (self.downsample[0].weight,) + self.downsample[1].args()
        else:
            ds = (None, None, None, None, None)
        return ops.BottleneckFn.apply(x, self.stride, self.training, self.conv1.weight, *self.bn1.args(),
                                      self.conv2.weight, *self.bn2.args(), self.conv3.weight, *self.bn3.args(), *ds, seg)


class ResNet18Trunk(nn.Module):
    """[F,3,H,W] NCHW frames -> NHWC feature map [F,H/32,W/32,512]."""

    def __init__(self):
        super().__init__()
        self.add_module("0", _Conv(3, 64, 7))
        self.add_module("1", _BN(64))
        self._build_stages()

    def _build_stages(self):
        for name, cin, cout, stride in STAGES:
            stage = nn.Sequential()
            stage.add_module("0", _Block(cin, cout, stride))
            stage.add_module("1", _Block(cout, cout, 1))
            self.add_module(name, stage)

    def forward(self, x):
        conv, bn = getattr(self, "0"), getattr(self, "1")
        ops.mark_cacheable(conv.weight)
        y = ops.StemFn.apply(x, conv.weight, *bn.args(), self.training)
        for name, _, _, _ in STAGES:
            if name == "7":
                y = _grad_ready(y)
            for blk in getattr(self, name):
                y = blk(y)
        if self.training and not ops.deferring():
            self.bump_counters(1)
        return y

    def counters(self):
        """{running_mean.data_ptr(): num_batches_tracked} of every BatchNorm (ops.apply_deferred bumps them in its own launch)"""
        return {m.running_mean.data_ptr(): m.num_batches_tracked for m in self.modules() if isinstance(m, _BN)}

    def bump_counters(self, n):
        """num_batches_tracked += n for every BatchNorm (one fused launch)"""
        bufs = [m.num_batches_tracked for m in self.modules() if isinstance(m, _BN)]
        torch._foreach_add_(bufs, n)


class ResNet50Trunk(ResNet18Trunk):
    """torchvision resnet50().children()[:-2] (resnet50_2fc.py:29-32): Bottleneck stages [3, 4, 6, 3], widths 64..512,
    output [F,H/32,W/32,2048]; 23 508 032 parameters."""

    def _build_stages(self):
        cin = 64
        for name, width, n, stride in (("4", 64, 3, 1), ("5", 128, 4, 2), ("6", 256, 6, 2), ("7", 512, 3, 2)):
            stage = nn.Sequential()
            for i in range(n):
                stage.add_module(str(i), _Bottleneck(cin, width, stride if i == 0 else 1))
                cin = width * 4
            self.add_module(name, stage)


# The two trunk calls of an episode (support frames, query frames: resnet18_2fc.py:41-42) are independent: they are
# queued on two HIP streams so that their kernels interleave on the GPU (the MFMA-bound convolutions of one call fill the
# workgroup-quantisation tails of the other and overlap its HBM-bound BatchNorm passes).  Each call still normalises with
# its OWN batch statistics; the running-statistics updates are deferred and applied in the reference's order
# (support, then query).  autograd runs each call's backward on the stream of its forward, so the backward overlaps too.
OVERLAP_TRUNK_CALLS = True
_BN_UPDATE_EVENT = {}


def _grad_ready(y):
    """the input of the last stage: when its gradient arrives, the backward pass has finished the last stage, the heads and the matcher
    (ops.GRAD_READY_HOOK = parallel.EarlyAllReduce.hook while an optimizer step is due)"""
    h = ops.GRAD_READY_HOOK
    if h is not None and torch.is_grad_enabled() and y.requires_grad:
        y.register_hook(h)
        owner = getattr(h, "__self__", None)
        if hasattr(owner, "registered"):
            owner.registered()
    return y

# Round 4: both trunk calls as ONE launch per layer.  The support and the query frames travel through the trunk as one NHWC tensor
# [Fs + Fq, H, W, C] with a frame split ("two frame segments", include/lmkd.h lmkd_*_seg): every convolution / BatchNorm / pooling kernel
# is launched once for both calls, each call still normalises with its OWN batch statistics ([2][5][C] tables; the row tiles of the
# convolutions are dealt per segment, so activations and activation gradients are bit-identical to two separate calls), weight and
# BatchNorm-parameter gradients are summed over both calls in one pass.  Half the trunk's launches (431 -> ~250 per episode), grids of
# twice the size, no second stream / shadow gradient buffer for the query call.  Exists wherever the arithmetic has the two-segment
# kernels (the bf16-plane modes: fp32-as-3xbf16 and bf16).
# OFF by default - measured, same box, alternating processes (profiles/r04_merge_ab.txt): 36.0 episodes/s merged against 36.8 on the
# two-call two-stream schedule (bf16 tensors: 70.7 against 75.7), although the serialized patch-kernel rate rises from 0.447 to 0.478 of its
# roofline and the host enqueues an episode in 7 ms instead of 11.  The kernel trace says why (profiles/r04_merge_trace.txt): no single
# kernel fills the chip (MFMA busy 54 - 62 %), and three streams of half-size launches co-schedule complementary kernels (one call's
# convolutions beside the other's BatchNorm passes and the weight gradients) all the time, while the merged chain has a partner only
# during the backward pass: its main stream is busy 96 % of the wall time - it IS the critical path - with the weight-gradient stream
# idle 42 %.  tools/tail_probe.py had predicted it per layer (profiles/r04_tail_probe.txt: one 400-frame launch = two concurrent
# 200-frame launches +- 3 %, layer 3 excepted).  For hosts short of cores (7 ms instead of 11 per episode) and as the base of a
# persistent-kernel schedule; tests/test_gpu_merged.py keeps it bit-identical to the two-call path.
MERGE_TRUNK_CALLS = False


def merge_supported(trunk, context_frames, target_frames):
    return (MERGE_TRUNK_CALLS and isinstance(trunk, ResNet18Trunk) and context_frames.is_cuda and ops.get_conv_compute_dtype() != "fp32"
            and context_frames.shape[1:] == target_frames.shape[1:] and context_frames.dtype == target_frames.dtype)


def merged_trunk_call(trunk, head, context_frames, target_frames):
    """-> head(trunk([context_frames; target_frames])) as ONE [Fs + Fq, ...] tensor plus Fs: the trunk runs once over both frame sets
    (resnet18_2fc.py:41-42 calls it twice; train-mode BatchNorm statistics stay per call)."""
    Fs = context_frames.shape[0]
    x4 = ops.frames_pair_to_nhwc4(context_frames, target_frames)
    training = trunk.training
    seg = Fs if training else 0      # eval: the running statistics serve every frame alike
    upd = []
    ops.mark_cacheable(getattr(trunk, "0").weight)
    try:
        if training:
            ops.set_defer(upd)
        y = ops.StemFn.apply(x4, getattr(trunk, "0").weight, *getattr(trunk, "1").args(), training, seg)
        for name, _, _, _ in STAGES:
            if name == "7":
                y = _grad_ready(y)
            for blk in getattr(trunk, name):
                y = blk(y, seg)
        feat = head(y)
    finally:
        ops.set_defer(None)
    if training:
        main = torch.cuda.current_stream()
        key = context_frames.device.index
        prev = _BN_UPDATE_EVENT.get(key)
        if prev is not None:
            main.wait_event(prev)
        # entries come in (support, query) pairs per BatchNorm: applied in the reference's order; num_batches_tracked += 2 in the same launch
        ops.apply_deferred(upd, trunk.counters())
        ev = torch.cuda.Event()
        ev.record(main)
        _BN_UPDATE_EVENT[key] = ev
    return feat, Fs


def trunk_features(trunk, head, context_frames, target_frames):
    """-> (X, Fs): X = head(trunk(.)) of the support frames (rows [0, Fs)) followed by the query frames' - from the merged trunk call
    where the arithmetic has it, else from the two-call schedule"""
    if merge_supported(trunk, context_frames, target_frames):
        return merged_trunk_call(trunk, head, context_frames, target_frames)
    cf, tf = two_trunk_calls(trunk, head, context_frames, target_frames)
    return torch.cat([cf, tf], 0), cf.shape[0]


def two_trunk_calls(trunk, head, context_frames, target_frames):
    """-> (head(trunk(context_frames)), head(trunk(target_frames))).  The two calls are enqueued layer by layer,
    alternating between the two streams, so both queues stay fed (and autograd, which walks the graph in reverse creation
    order, alternates between them in the backward as well)."""
    if merge_supported(trunk, context_frames, target_frames):
        feat, Fs = merged_trunk_call(trunk, head, context_frames, target_frames)
        return feat[:Fs], feat[Fs:]
    if not (OVERLAP_TRUNK_CALLS and context_frames.is_cuda):
        return head(trunk(context_frames)), head(trunk(target_frames))
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    main = torch.cuda.current_stream()
    side = ops.side_stream(context_frames.device)
    side.wait_stream(main)
    target_frames.record_stream(side)
    q_upd, s_upd = [], []
    defer = trunk.training
    ops.mark_cacheable(getattr(trunk, "0").weight)
    layers = [lambda t: ops.StemFn.apply(t, getattr(trunk, "0").weight, *getattr(trunk, "1").args(), trunk.training)]
    for name, _, _, _ in STAGES:
        if name == "7":
            layers.append(_grad_ready)
        layers += list(getattr(trunk, name))
    layers.append(head)
    cf, tf = context_frames, target_frames
    try:
        for layer in layers:
            if defer:
                ops.set_defer(s_upd)
            cf = layer(cf)
            with torch.cuda.stream(side):
                if defer:
                    ops.set_defer(q_upd)
                tf = layer(tf)
    finally:
        ops.set_defer(None)
    main.wait_stream(side)
    tf.record_stream(main)
    if defer:
        # the running-statistics updates of successive episodes stay in program order also when the episodes' forwards run on
        # different stream sets (trainloop.PipelinedEpisodes)
        key = context_frames.device.index
        prev = _BN_UPDATE_EVENT.get(key)
        if prev is not None:
            main.wait_event(prev)
        ops.apply_deferred(s_upd + q_upd, trunk.counters())
        ev = torch.cuda.Event()
        ev.record(main)
        _BN_UPDATE_EVENT[key] = ev
    return cf, tf


class Linear(nn.Linear):
    """nn.Linear parameters/init, forward on the MFMA GEMM."""

    def forward(self, x):
        return ops.LinearFn.apply(x, self.weight, self.bias)
