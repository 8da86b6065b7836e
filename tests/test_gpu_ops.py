"""GPU parity tests of every HIP op against the CPU oracle (torch CPU fp32 / oracle.ref_cpu).
All calls go through the C ABI (lite-mkd_amd/_lib.py -> liblmkd_hip.so).  Tolerances are fp32:
different summation order vs the CPU library, stated per test."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import litemkd_amd
    litemkd_amd.lib().call("lmkd_device_check", 0)
    return torch.device("cuda", 0)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(a, b, rtol, atol, msg="", flip_frac=0.0):
    """elementwise |a-b| <= atol + rtol*|b|.  flip_frac > 0 switches to a relative-L2 criterion (< 2e-2) for gradients
    that pass through ReLU masks: a pre-activation within fp32 rounding of zero flips its mask between the two
    implementations and perturbs one 3x3xC neighbourhood of the gradient, so a few elements differ by percents."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if flip_frac > 0.0:
        l2 = float((a - b).norm() / (b.norm() + 1e-30))
        assert l2 < 2e-2, "%s: %d/%d elements differ, rel-L2 %.3e" % (msg, int(bad.sum()), bad.numel(), l2)
        return
    assert not bad.any(), "%s: %d/%d elements differ, max abs err %.3e (ref max %.3e)" % (
        msg, int(bad.sum()), bad.numel(), float(err.max()), float(b.abs().max()))


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layA,layB", [("K", "K"), ("K", "N"), ("M", "K"), ("M", "N")])
@pytest.mark.parametrize("M,N,K", [(400, 4608, 2048), (700, 700, 1152), (28, 36, 140), (132, 260, 64)])
def test_gemm(dev, layA, layB, M, N, K):
    from litemkd_amd import ops
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)
    bias = rnd(N, seed=3)
    ref = 0.5 * (A.double() @ B.double()).float() + bias
    Ad = (A if layA == "K" else A.t()).contiguous().to(dev)
    Bd = (B.t() if layB == "K" else B).contiguous().to(dev)
    C = torch.empty(M, N, device=dev)
    ops.gemm(layA, layB, M, N, K, Ad, Ad.shape[1], Bd, Bd.shape[1], C, N, alpha=0.5, bias=bias.to(dev))
    close(C, ref, 1e-4, 1e-4 * math.sqrt(K), "gemm %s%s" % (layA, layB))


def test_gemm_batched_beta_relu(dev):
    from litemkd_amd import ops
    A = rnd(3, 64, 96, seed=4)
    B = rnd(3, 80, 96, seed=5)
    C0 = rnd(3, 64, 80, seed=6)
    ref = torch.relu(torch.einsum("bmk,bnk->bmn", A, B) + 2.0 * C0)
    C = C0.clone().to(dev)
    ops.gemm("K", "K", 64, 80, 96, A.to(dev), 96, B.to(dev), 96, C, 80, beta=2.0, relu=True, batch=3,
             sA=64 * 96, sB=80 * 96, sC=64 * 80)
    close(C, ref, 1e-4, 1e-3, "batched gemm")


@pytest.mark.parametrize("layA,layB", [("K", "K"), ("K", "N"), ("M", "N")])
@pytest.mark.parametrize("M,N,K,batch", [(200, 512, 2048, 1), (200, 512, 8192, 1), (100, 700, 1152, 1), (96, 64, 4096, 3)])
def test_gemm_split_k(dev, layA, layB, M, N, K, batch):
    """lmkd_gemm_f32_splitk (the K range over several workgroups, partial tiles added in split order by the last-arriving block):
    against fp64, against the unsplit kernel, bit-identical from run to run, and the ticket words are zero again afterwards"""
    from litemkd_amd import ops
    A = rnd(batch, M, K, seed=7)
    B = rnd(batch, K, N, seed=8)
    C0 = rnd(batch, M, N, seed=9)
    bias = rnd(N, seed=10)
    ref = torch.relu(0.25 * torch.einsum("bmk,bkn->bmn", A.double(), B.double()) + 0.5 * C0.double() + bias.double()).float()
    Ad = (A if layA == "K" else A.transpose(1, 2)).contiguous().to(dev)
    Bd = (B.transpose(1, 2) if layB == "K" else B).contiguous().to(dev)

    def run(split):
        was = ops.GEMM_SPLIT_K
        ops.GEMM_SPLIT_K = split
        try:
            C = C0.clone().to(dev)
            ops.gemm(layA, layB, M, N, K, Ad, Ad.shape[2], Bd, Bd.shape[2], C, N, alpha=0.25, beta=0.5, bias=bias.to(dev), relu=True,
                     batch=batch, sA=M * K, sB=K * N, sC=M * N)
            return C
        finally:
            ops.GEMM_SPLIT_K = was
    c1, c2, c0 = run(True), run(True), run(False)
    assert torch.equal(c1, c2)
    assert not torch.equal(c1, c0)          # the split did happen (another summation order)
    close(c1, ref, 1e-4, 1e-4 * math.sqrt(K), "split-K gemm %s%s" % (layA, layB))
    close(c1, c0, 1e-5, 1e-5 * math.sqrt(K), "split-K vs unsplit")
    ws, tk = ops._gemm_workspace(c1)
    assert int(tk.abs().sum()) == 0


def test_gemm_rejects_bad_alignment(dev):
    from litemkd_amd import ops
    A = torch.zeros(8, 6, device=dev)
    with pytest.raises(RuntimeError):
        ops.gemm("K", "K", 8, 8, 6, A, 6, A, 6, torch.zeros(8, 8, device=dev), 8)


# ------------------------------------------------------------------------------------------
CONVS = [  # N, Cin, H, W, Cout, K, stride, pad
    (3, 3, 64, 64, 64, 7, 2, 3),       # stem (NHWC4 path)
    (2, 64, 28, 28, 64, 3, 1, 1),      # layer1
    (2, 64, 28, 28, 128, 3, 2, 1),     # layerN.0.conv1
    (2, 128, 14, 14, 128, 3, 1, 1),
    (2, 64, 28, 28, 128, 1, 2, 0),     # downsample
    (5, 256, 7, 7, 512, 3, 1, 1),      # ragged row tile (M = 245)
    (3, 64, 7, 7, 128, 3, 2, 1),       # stride 2 on an odd map (7 -> 4): parity classes of unequal size
    (3, 64, 9, 5, 64, 1, 2, 0),        # 1x1 stride 2, odd H and W
    (3, 3, 100, 84, 64, 7, 2, 3),      # stem sizes of stem_wgrad_kernel: 3 steps per unit (dy fetched 2 steps ahead)
    (2, 3, 97, 150, 64, 7, 2, 3),      # odd output height (the last unit of an image has one row), 5 steps
    (5, 3, 224, 224, 64, 7, 2, 3),     # 7 steps per unit (6 ahead), several units per slab
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_bwd(dev, cfg):
    from litemkd_amd import ops
    N, Cin, H, W, Cout, K, s, p = cfg
    x = rnd(N, Cin, H, W, seed=10).requires_grad_()
    w = (rnd(Cout, Cin, K, K, seed=11) * math.sqrt(2.0 / (Cout * K * K))).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    gy = rnd(*y.shape, seed=12)
    y.backward(gy)
    Cs = 4 if Cin == 3 else Cin
    xd = torch.zeros(N, H, W, Cs)
    xd[..., :Cin] = nhwc(x.detach())
    xd, wdv = xd.to(dev), w.detach().to(dev)
    wp = ops.pack_weights(wdv, Cs, 0)
    yd, part = ops.conv_fwd(xd, wp, Cout, K, K, s, p, True)
    tol = 2e-5 * math.sqrt(Cin * K * K)
    close(nchw(yd), y, 1e-4, tol, "conv fwd")
    # fused BN statistics: per-channel sum and sum of squares
    sums = part.double().sum(0).cpu()
    close(sums[:, 0], y.detach().double().sum((0, 2, 3)), 1e-4, 1e-2, "bn sum")
    close(sums[:, 1], (y.detach().double() ** 2).sum((0, 2, 3)), 1e-4, 1e-2, "bn sumsq")
    gyd = nhwc(gy).to(dev)
    dw = ops.conv_bwd_weight(xd, gyd, tuple(w.shape), s, p)
    close(dw, w.grad, 1e-3, 2e-5 * math.sqrt(N * y.shape[2] * y.shape[3]) * 3, "conv wgrad")
    if Cin != 3:
        wd = ops.pack_weights(wdv, Cin, 1)
        dx = ops.conv_bwd_data(gyd, wd, (N, H, W, Cin), Cout, K, K, s, p)
        close(nchw(dx), x.grad, 1e-4, 2e-5 * math.sqrt(Cout * K * K), "conv dgrad")


def test_bn_train_apply_backward(dev):
    from litemkd_amd import ops
    N, C, H, W = 4, 64, 14, 14
    x = (rnd(N, C, H, W, seed=20) * 2 + 0.5).requires_grad_()
    res = rnd(N, C, H, W, seed=21).requires_grad_()
    gamma = (1 + 0.1 * rnd(C, seed=22)).requires_grad_()
    beta = (0.1 * rnd(C, seed=23)).requires_grad_()
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5) + res)
    gy = rnd(N, C, H, W, seed=24)
    y.backward(gy)
    xd = nhwc(x.detach()).to(dev)
    # statistics from the (sum, sumsq) partial layout the conv epilogue produces
    flat = xd.reshape(-1, C)
    T = 7
    chunks = flat.reshape(T, -1, C)
    part = torch.stack([chunks.sum(1), (chunks ** 2).sum(1)], -1).contiguous()
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    st = ops.bn_stats_train(part, flat.shape[0], gamma.detach().to(dev), beta.detach().to(dev), rmd, rvd)
    close(rmd, rm, 1e-4, 1e-5, "running_mean")
    close(rvd, rv, 1e-4, 1e-5, "running_var")
    yd = ops.bn_apply(xd, st, True, nhwc(res.detach()).to(dev))
    close(nchw(yd), y, 1e-4, 1e-4, "bn apply")
    dx, g, dg, db = ops.bn_backward(nhwc(gy).to(dev), xd, yd, st, gamma.detach().to(dev), 1, want_g=True)
    close(nchw(dx), x.grad, 1e-3, 2e-5, "bn dx")
    close(nchw(g), res.grad, 1e-5, 1e-6, "residual grad")
    close(dg, gamma.grad, 1e-3, 1e-3, "dgamma")
    close(db, beta.grad, 1e-3, 1e-3, "dbeta")


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "bf16"])
@pytest.mark.parametrize("cfg", [(5, 64, 14, 14, 64, 3, 1, 1), (3, 128, 9, 11, 128, 3, 2, 1), (4, 256, 7, 7, 64, 1, 1, 0), (40, 128, 28, 28, 128, 3, 1, 1)])
def test_conv_fused_bn_relu_loader_bit_identical(dev, cfg, mode):
    """training-mode fusion (lmkd_conv2d_fwd_pre / lmkd_conv2d_bwd_weight_pre): the convolution reads the RAW previous conv
    output and applies relu(BatchNorm(.)) in its loader; results (output, BatchNorm partial sums, weight gradient) must be
    BIT-identical to materialising the activation with lmkd_bn_apply first — zero padding included (it pads the activation)."""
    from litemkd_amd import ops
    N, C, H, W, Cout, K, s, p = cfg
    c1 = (rnd(N, H, W, C, seed=40) * 1.5 + 0.2).to(dev)
    gamma, beta = (1 + 0.2 * rnd(C, seed=41)).to(dev), (0.3 * rnd(C, seed=42)).to(dev)
    flat = c1.reshape(-1, C)
    part = torch.stack([flat.sum(0, keepdim=True), (flat ** 2).sum(0, keepdim=True)], -1).contiguous()
    st = ops.bn_stats_train(part, flat.shape[0], gamma, beta, None, None)
    w = (rnd(Cout, C, K, K, seed=43) * math.sqrt(2.0 / (Cout * K * K))).to(dev)
    ops.set_conv_compute_dtype(mode)
    try:
        wp = ops.pack_weights(w, C, 0)
        a1 = ops.bn_apply(c1, st, True)
        y0, p0 = ops.conv_fwd(a1, wp, Cout, K, K, s, p, True)
        y1, p1 = ops.conv_fwd(c1, wp, Cout, K, K, s, p, True, pre_stats=st)
        assert torch.equal(y0, y1) and torch.equal(p0, p1)
        dy = rnd(*y0.shape, seed=44).to(dev)
        dw0 = ops.conv_bwd_weight(a1, dy, (Cout, C, K, K), s, p)
        dw1 = ops.conv_bwd_weight(c1, dy, (Cout, C, K, K), s, p, pre_stats=st)
        assert torch.equal(dw0, dw1)
    finally:
        ops.reset_compute_dtypes()
    assert float((a1 == 0).float().mean()) > 0.2          # the ReLU did clip: the test would not see a missing max otherwise


@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9", "fp32", "bf16"])
@pytest.mark.parametrize("cfg", [(5, 64, 14, 14, 64, 3), (3, 128, 9, 11, 64, 3), (4, 256, 7, 7, 128, 1), (40, 128, 28, 28, 128, 3), (60, 64, 56, 56, 64, 3)])
def test_dgrad_epilogue_bn_backward_sums(dev, cfg, mode):
    """lmkd_conv2d_bwd_data_bn + lmkd_bn_backward_part (the BatchNorm backward's reduction in the data gradient's epilogue) against
    lmkd_conv2d_bwd_data + lmkd_bn_backward: the data gradient bit-identical, the BatchNorm backward equal up to the order of the
    fp32 partial sums (fp64 from the row tiles on), and both against an fp64 reference of the sums.  Modes without the fused form
    (native fp32, one-plane bf16) must report 0 tiles and take the plain path."""
    from litemkd_amd import ops
    N, Cout, H, W, Cin, K = cfg
    pad = K // 2
    c1 = (rnd(N, H, W, Cin, seed=50) * 1.5 + 0.2).to(dev)
    gamma, beta = (1 + 0.2 * rnd(Cin, seed=51)).to(dev), (0.3 * rnd(Cin, seed=52)).to(dev)
    flat = c1.reshape(-1, Cin)
    part = torch.stack([flat.sum(0, keepdim=True), (flat ** 2).sum(0, keepdim=True)], -1).contiguous()
    st = ops.bn_stats_train(part, flat.shape[0], gamma, beta, None, None)
    w = (rnd(Cout, Cin, K, K, seed=53) * math.sqrt(2.0 / (Cout * K * K))).to(dev)
    dy = rnd(N, H, W, Cout, seed=54).to(dev)
    ops.set_conv_compute_dtype(mode)
    was = ops.DGRAD_BN_STATS
    ops.DGRAD_BN_STATS = True      # (the default)
    try:
        wd = ops.pack_weights(w, Cin, 1)
        T = ops.lib().value("lmkd_conv2d_bwd_data_bn_tiles", N, H, W, Cin, Cout, K, K, 1, pad)
        assert (T > 0) == (mode in ("fp32x3", "fp32x3_9")), T
        da0 = ops.conv_bwd_data(dy, wd, c1.shape, Cout, K, K, 1, pad)
        da1, p1 = ops.conv_bwd_data(dy, wd, c1.shape, Cout, K, K, 1, pad, bn=(c1, st))
        assert torch.equal(da0, da1)
        assert (p1 is not None) == (T > 0)
        dc0, _, dg0, db0 = ops.bn_backward(da0.clone(), c1, None, st, gamma, 2)
        dc1, _, dg1, db1 = ops.bn_backward(da1.clone(), c1, None, st, gamma, 2, part=p1)
    finally:
        ops.DGRAD_BN_STATS = was
        ops.reset_compute_dtypes()
    if p1 is None:
        assert torch.equal(dc0, dc1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
        return
    assert p1.shape == (T, Cin, 2)
    g = (da0 * (torch.addcmul(st[3], c1, st[2]) > 0)).double().reshape(-1, Cin)
    xh = ((c1 - st[0]) * st[1]).double().reshape(-1, Cin)
    sg, sgx = g.sum(0), (g * xh).sum(0)
    scale = float(g.abs().sum(0).max())
    for name, got in (("fused", (db1, dg1)), ("plain", (db0, dg0))):
        assert float((got[0].double() - sg).abs().max()) <= 2e-6 * scale, name
        assert float((got[1].double() - sgx).abs().max()) <= 2e-6 * scale * float(xh.abs().max()), name
    assert float((dc1 - dc0).abs().max()) <= 1e-5 * float(dc0.abs().max())
    assert float((g != 0).double().mean()) < 0.9          # the mask did clip


def test_bn_apply_relu_bit_mask(dev):
    """lmkd_bn_apply's packed ReLU mask (bit e = y[e] > 0) and lmkd_bn_backward(mask_mode 3) reading it: bit-identical to the
    backward that reads y itself (mask_mode 1)"""
    from litemkd_amd import ops
    N, C, H, W = 6, 128, 9, 7
    x = (rnd(N, H, W, C, seed=50) * 2).to(dev)
    res = rnd(N, H, W, C, seed=51).to(dev)
    gamma, beta = (1 + 0.1 * rnd(C, seed=52)).to(dev), (0.1 * rnd(C, seed=53)).to(dev)
    flat = x.reshape(-1, C)
    part = torch.stack([flat.sum(0, keepdim=True), (flat ** 2).sum(0, keepdim=True)], -1).contiguous()
    st = ops.bn_stats_train(part, flat.shape[0], gamma, beta, None, None)
    y, bits = ops.bn_apply(x, st, True, res, want_bits=True)
    assert torch.equal(y, ops.bn_apply(x, st, True, res))
    un = ((bits.reshape(-1, 1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1).reshape(-1).bool()
    assert torch.equal(un, (y > 0).reshape(-1))
    dy = rnd(N, H, W, C, seed=54).to(dev)
    r1 = ops.bn_backward(dy, x, y, st, gamma, 1, want_g=True)
    r3 = ops.bn_backward(dy, x, bits, st, gamma, 3, want_g=True)
    for a, b in zip(r1, r3):
        assert torch.equal(a, b)


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "bf16"])
@pytest.mark.parametrize("kind,cin,cout,stride", [("basic", 64, 128, 2), ("basic", 64, 64, 1), ("bottleneck", 256, 128, 2), ("bottleneck", 512, 128, 1)])
def test_block_fused_training_path_bit_identical(dev, kind, cin, cout, stride, mode):
    """BasicBlock / Bottleneck with the BatchNorm+ReLU of the inner activations fused into the consumers' loaders and the
    output mask as bits (ops.FUSE_TRAIN_BN) vs the materialising path: outputs, input gradient and every parameter gradient
    bit-identical (same arithmetic, fewer passes over HBM)"""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone import resnet as R
    torch.manual_seed(3)
    blk = (R._Block(cin, cout, stride) if kind == "basic" else R._Bottleneck(cin, cout, stride)).to(dev).train()
    with torch.no_grad():
        for n_, p_ in blk.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    x = torch.relu(torch.randn(6, 12, 12, cin, device=dev))
    gy = None
    outs = []
    ops.set_conv_compute_dtype(mode)
    ops.FUSE_PRE_ALL_MODES = True          # the loaders of the bf16-plane kernels too (policy: native fp32 only, ops._train_pre)
    # the Bottleneck's stride-2 3x3 convolution: with the loader fusion it runs the gather kernel (the patch kernel's stride-2 form has
    # no BatchNorm loader), without it the patch form, whose taps are summed class by class - same kernel for both arms here
    ops.lib().call("lmkd_conv_set_s2_patch", 0)
    for fuse in (False, True):
        ops.FUSE_TRAIN_BN = fuse
        try:
            blk.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_()
            y = blk(xi)
            if gy is None:
                gy = torch.randn_like(y)
            y.backward(gy)
            outs.append((y.detach().clone(), xi.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()}))
        finally:
            ops.FUSE_TRAIN_BN = True
            if fuse:
                ops.reset_compute_dtypes()
                ops.FUSE_PRE_ALL_MODES = False
    (y0, dx0, g0), (y1, dx1, g1) = outs
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def _trunk_params(seed):
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(seed)
    sd = O.init_resnet18_trunk(g)
    # non-trivial BN affine so gamma/beta paths are exercised
    for k in list(sd):
        if k.endswith(".weight") and sd[k].dim() == 1:
            sd[k] = 1 + 0.1 * torch.randn(sd[k].shape, generator=g)
        if k.endswith(".bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


@pytest.mark.parametrize("mode", ["fp32x3", "fp32"])
def test_trunk_forward_backward(dev, mode):
    """whole ResNet-18 trunk (stem + 8 BasicBlocks + pooled head) fwd + bwd vs the oracle, 6 frames 64x64.  Gradients are
    judged against an fp64 run of the oracle: per tensor, the HIP error may be at most 3x the error of the oracle's own fp32
    run (tests/_anchor.py) — ReLU-mask flips at pre-activations within rounding of zero show on both fp32 sides.
    mode: library default (fp32x3, the benchmark's arithmetic) and the native fp32 MFMA."""
    from litemkd_amd import ops
    ops.set_conv_compute_dtype(mode)
    from litemkd_amd.model.backbone.resnet import ResNet18Trunk
    from oracle import ref_cpu as O
    from _anchor import anchored_dict
    sd = _trunk_params(5)
    trunk = ResNet18Trunk()
    trunk.load_state_dict(sd)
    trunk = trunk.to(dev).train()
    x = torch.rand(6, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    gfeat = rnd(6, 512, seed=7)

    def oracle(dt):
        osd = {k: (v.clone().to(dt).requires_grad_() if v.is_floating_point() and "running" not in k else
                   (v.clone().to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
        fm = O.resnet18_trunk(x.to(dt), osd, True, True, {})
        feat = O.pooled_frame_features(fm)
        feat.backward(gfeat.to(dt))
        return osd, fm, feat
    osd, fm, feat = oracle(torch.float32)
    osd64, fm64, feat64 = oracle(torch.float64)
    y = trunk(x.to(dev))
    close(nchw(y), fm, 2e-3, 2e-3, "trunk feature map")
    f = ops.PoolHeadFn.apply(y)
    close(f, feat, 2e-3, 2e-3, "pooled features")
    f.backward(gfeat.to(dev))
    close(trunk.state_dict()["1.running_var"], osd["1.running_var"], 1e-3, 1e-4, "stem running_var")
    close(trunk.state_dict()["7.1.bn2.running_mean"], osd["7.1.bn2.running_mean"], 1e-3, 1e-4, "last running_mean")
    names = [k for k, _ in trunk.named_parameters()]
    worst = anchored_dict({k: p.grad for k, p in trunk.named_parameters()}, {k: osd[k].grad for k in names},
                          {k: osd64[k].grad for k in names})
    print("trunk gradients, worst HIP/CPU error ratio vs fp64:", worst)
    from _anchor import record
    record("trunk 6 frames 64px [%s]" % mode, {"worst parameter gradient (%s)" % worst[1]: worst[2:]})


class _MaskedReLU(torch.autograd.Function):
    """ReLU with an imposed 0/1 mask (forward t * mask, backward g * mask)"""

    @staticmethod
    def forward(ctx, t, mask):
        ctx.save_for_backward(mask)
        return t * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


BLOCK_CASES = [(6, 256, 512, 6, 2), (6, 512, 512, 3, 1), (6, 64, 128, 24, 2), (6, 64, 64, 24, 1),
               (40, 64, 128, 56, 2), (40, 128, 128, 28, 1), (40, 256, 512, 14, 2),
               (200, 128, 256, 28, 2), (200, 256, 256, 14, 1)]
# every case in the library default (= the benchmark's headline arithmetic, fp32x3) and in the native fp32 MFMA mode; the 40- and
# 200-frame cases also with bf16 tensors in HBM (BASELINE configs[2]) against references with the same rounding points
BLOCK_PARAMS = [c + (m,) for c in BLOCK_CASES for m in ("fp32x3", "fp32")] + [c + ("bf16act",) for c in BLOCK_CASES if c[0] >= 40]


@pytest.mark.parametrize("N,cin,cout,H,stride,mode", BLOCK_PARAMS)
def test_block_isolated(dev, N, cin, cout, H, stride, mode):
    """one BasicBlock (with / without downsample) fwd + bwd on identical inputs.  The hand-scheduled backward (accumulate
    epilogue, bn1 mask recompute, strided 1x1 gradient accumulated onto the pixels it reaches) is judged per tensor against an
    fp64 evaluation: error at most 3x torch-CPU-fp32's own error vs fp64 (tests/_anchor.py).

    Two fp32 evaluations of a ReLU network legitimately differ where a pre-activation lies within rounding of zero: ONE flipped
    mask among 4 M elements moves a channel's gradient sum by ~1e-3 of the tensor norm (measured: the 40-frame cases flip 1-2
    elements on the HIP side, none on the CPU side).  So the references run with the HIP path's OWN masks imposed (both ReLUs),
    after checking that those masks differ from the fp64 masks only where the fp64 pre-activation is within 1e-5 of zero: what
    remains is a linear map of the upstream gradient, where any indexing / scheduling / accumulation bug shows at full size.
    The 40-frame cases are large enough for the 128x128 tile, the 4-class stride-2 data gradient with many tiles per class,
    XCD-banded tile orders and multi-round weight-gradient splits; the 200-frame cases ARE two of the benchmark's blocks (layer 3,
    with and without downsample): full-size backward with the accumulate epilogue, the strided 1x1 gradient accumulating onto
    the pixels it reaches and the BatchNorm passes, all inside one hand-scheduled backward.

    mode: the arithmetic of the convolutions - fp32x3 (library default = what bench.py's headline line runs: patch / window /
    gather plane kernels), fp32 (native fp32 MFMA, fused BatchNorm loaders), bf16act (bf16 tensors in HBM, one-plane kernels):
    there the references are evaluated with the oracle's rounding points (oracle.CONV_BF16: every convolution GEMM on bf16-rounded
    operands; oracle.ACT_BF16: every stored activation / gradient rounded), in fp32 and in fp64."""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone import resnet as R
    from oracle import ref_cpu as O
    from _anchor import anchored, anchored_dict, record
    b16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if b16 else mode)
    ops.set_activation_dtype("bf16" if b16 else "fp32")
    torch.manual_seed(0)
    blk = R._Block(cin, cout, stride)
    x0 = torch.relu(torch.randn(N, cin, H, H))
    gy = torch.randn(N, cout, H // stride, H // stride)
    if b16:      # the block's input and the upstream gradient are stored tensors: bf16 values
        x0, gy = x0.bfloat16().float(), gy.bfloat16().float()
    params = {k: v.detach().clone() for k, v in blk.named_parameters()}
    blk = blk.to(dev).train()
    act = torch.bfloat16 if b16 else torch.float32
    xd = nhwc(x0).to(dev).to(act).requires_grad_()
    ops.BLOCK_TAPS = []
    try:
        yd = blk(xd)
        tap = ops.BLOCK_TAPS[0]
    finally:
        ops.BLOCK_TAPS = None
    yd.backward(nhwc(gy).to(dev).to(act))
    m1 = nchw(ops.bn_apply(tap["c1"], tap["st1"], True) > 0).cpu()          # fmaf(c1, scale, shift) > 0: what the backward recomputes
    my = nchw(yd.detach() > 0).cpu()

    def ref_run(dt, masks):
        x = x0.detach().clone().to(dt).requires_grad_()
        ref = {k: v.detach().clone().to(dt).requires_grad_() for k, v in params.items()}

        def bn(t, pre):
            return F.batch_norm(t, torch.zeros(cout, dtype=dt), torch.ones(cout, dtype=dt), ref[pre + ".weight"], ref[pre + ".bias"],
                                True, 0.1, 1e-5)

        def relu(t, i):
            return F.relu(t) if masks is None else _MaskedReLU.apply(t, masks[i].to(dt))
        O.CONV_BF16, O.ACT_BF16 = b16, b16
        try:      # the rounding points of oracle.resnet18_trunk's block body (ref_cpu.py:183-190)
            pre1 = bn(O._act(O._conv(x, ref["conv1.weight"], stride, 1)), "bn1")
            out = bn(O._act(O._conv(O._act(relu(pre1, 0)), ref["conv2.weight"], 1, 1)), "bn2")
            idn = x
            if "downsample.0.weight" in ref:
                idn = bn(O._act(O._conv(x, ref["downsample.0.weight"], stride, 0)), "downsample.1")
            pre2 = out + idn
            y = O._act(relu(pre2, 1))
            y.backward(gy.to(dt))
        finally:
            O.CONV_BF16, O.ACT_BF16 = False, False
        g = {k: v.grad for k, v in ref.items()}
        # bf16 tensors: the block-input gradient is itself a stored tensor (the previous block's output gradient): the fp32 run
        # rounds it where the HIP path does, the fp64 run stays exact
        g["x"] = O._r16(x.grad) if (b16 and dt == torch.float32) else x.grad
        return y.detach(), g, (pre1.detach(), pre2.detach())
    y64n, _, (p1, p2) = ref_run(torch.float64, None)
    flips = 0
    for m, pre in ((m1, p1), (my, p2)):
        diff = m != (pre > 0)
        flips += int(diff.sum())
        # a flipped mask is legitimate only at a pre-activation within rounding of zero (fp32 rounding; with bf16 tensors an
        # activation within fp32 rounding of a bf16 rounding boundary moves by a bf16 ulp = 0.4 %)
        lim = (3e-2 if b16 else 1e-5) * float(pre.abs().max())
        cap = max(4, m.numel() // (500 if b16 else 200000))
        assert int(diff.sum()) <= cap and (not bool(diff.any()) or float(pre[diff].abs().max()) < lim), \
            "ReLU masks of the HIP path differ from fp64 at %d elements (|pre| up to %.2e)" % (int(diff.sum()), float(pre[diff].abs().max()))
    y32, g32, _ = ref_run(torch.float32, (m1, my))
    y64, g64, _ = ref_run(torch.float64, (m1, my))
    floor = 5e-5 if b16 else 2e-6      # bf16 tensors: one element rounding the other way moves it by 0.4 % in either fp32 evaluation
    ey = anchored("block y", nchw(yd.float()), y32, y64, 3.0, floor)
    hip = {k: v.grad for k, v in blk.named_parameters()}
    hip["x"] = nchw(xd.grad.float())
    worst = anchored_dict(hip, g32, g64, 3.0, floor)
    record("block N=%d %d->%d H=%d s=%d [%s]" % (N, cin, cout, H, stride, mode), {"y": ey, "worst grad (%s)" % worst[1]: worst[2:]})
    print("block gradients (HIP masks imposed, %d flips vs fp64), worst HIP/CPU error ratio vs fp64:" % flips, worst)


def test_pool_head(dev):
    from litemkd_amd import ops
    x = rnd(5, 512, 7, 7, seed=30).requires_grad_()
    from oracle import ref_cpu as O
    y = O.pooled_frame_features(x)
    gy = rnd(*y.shape, seed=31)
    y.backward(gy)
    xd = nhwc(x.detach()).to(dev).requires_grad_()
    yd = ops.PoolHeadFn.apply(xd)
    close(yd, y, 1e-6, 1e-6, "adaptive max+mean")
    yd.backward(gy.to(dev))
    close(nchw(xd.grad), x.grad, 1e-6, 1e-7, "adaptive max+mean bwd")


def test_linear(dev):
    from litemkd_amd import ops
    x = rnd(200, 512, seed=40).requires_grad_()
    w = (rnd(2048, 512, seed=41) * 0.05).requires_grad_()
    b = rnd(2048, seed=42).requires_grad_()
    y = F.linear(x, w, b)
    gy = rnd(200, 2048, seed=43)
    y.backward(gy)
    xd, wd, bd = (t.detach().to(dev).requires_grad_() for t in (x, w, b))
    yd = ops.LinearFn.apply(xd, wd, bd)
    yd.backward(gy.to(dev))
    close(yd, y, 1e-4, 1e-4, "linear")
    close(xd.grad, x.grad, 1e-4, 1e-4, "linear dx")
    close(wd.grad, w.grad, 1e-4, 1e-3, "linear dw")
    close(bd.grad, b.grad, 1e-4, 1e-3, "linear db")


# ------------------------------------------------------------------------------------------
def _args(dev, **kw):
    from litemkd_amd.options import default_args
    return default_args(device=dev, trans_dropout=0.0, **kw)


def _gsum(g):
    return g.reshape(g.shape[0], 8, 32, 64).sum(-1)


@pytest.mark.parametrize("case", [0, 1])
def test_head_golden(dev, golden_dir, case):
    """A3 + A1 against the REFERENCE: tests/golden/head.npz holds what the reference's own resnet18_2fc.forward
    (model/backbone/resnet18_2fc.py:44-77) and Student.forward (model/model_select.py:26-36) return for seeded trunk output
    maps (the absent torchvision trunk stubbed by an identity: oracle/gen_golden.py gen_head).  The HIP student with its trunk
    replaced by the same identity (NCHW -> NHWC, the layout the HIP trunk emits) - PoolHeadFn, LinearFn, the TRX heads, SupportDK and
    the dict plumbing - must reproduce features, logits and the gradients w.r.t. the maps and the fc parameters."""
    import os
    import torch.nn as nn
    from litemkd_amd.model.model_select import Student
    from litemkd_amd.model.backbone import resnet as R
    from oracle.gen_golden import head_case_inputs
    G = np.load(os.path.join(golden_dir, "head.npz"))
    pre = "c%d_" % case
    ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
    fm_s, fm_q, lab, params = head_case_inputs(int(G[pre + "seed"]), ns, nq, int(G[pre + "hw"]))
    args = _args(dev, shot=ns // 5, query_per_class=nq // 5)
    student = Student(args)

    class _ToNHWC(nn.Module):
        def forward(self, x):
            return x.permute(0, 2, 3, 1).contiguous()
    student.backbone.resnet = _ToNHWC()
    student = student.to(dev)
    sd = student.state_dict()
    for k, v in params.items():
        sd[k].copy_(v)
    fs, fq = fm_s.to(dev).requires_grad_(), fm_q.to(dev).requires_grad_()
    overlap = R.OVERLAP_TRUNK_CALLS
    R.OVERLAP_TRUNK_CALLS = False
    try:
        r = student(fs, lab.to(dev), fq)
    finally:
        R.OVERLAP_TRUNK_CALLS = overlap
    cf, tf, lg = r["context_features"], r["target_features"], r["logits"]
    assert sorted(r.keys()) + sorted(cf.keys()) + sorted(tf.keys()) + sorted(lg.keys()) == list(G[pre + "keys"])
    for k, t in (("cf1", cf["context_features_1"]), ("cf2", cf["context_features_2"]), ("tf1", tf["target_features_1"]),
                 ("tf2", tf["target_features_2"])):
        close(_gsum(t), torch.from_numpy(G[pre + k]), 1e-4, 1e-4, k)
    close(cf["context_features_1"][0], torch.from_numpy(G[pre + "cf1_v0"]), 1e-4, 2e-5, "cf1_v0")
    close(tf["target_features_2"][0], torch.from_numpy(G[pre + "tf2_v0"]), 1e-4, 2e-5, "tf2_v0")
    for k in ("kl", "ce", "sup"):
        close(lg[k], torch.from_numpy(G[pre + k]), 2e-4, 2e-2, k)
    w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5).to(dev)
    wf = torch.linspace(-1, 1, 2048).to(dev)
    ((lg["kl"] * w).sum() * 1e-2 + (lg["ce"] * w.flip(0)).sum() * 1e-2 + (lg["sup"] * torch.linspace(1, -1, 20).reshape(5, 4).to(dev)).sum() * 1e-3
     + (cf["context_features_1"] * wf).sum() * 1e-3 + (tf["target_features_2"] * wf.flip(0)).sum() * 1e-3).backward()
    for name, t in (("g_fm_s", fs.grad), ("g_fm_q", fq.grad)):
        ref_fc, ref_f0 = torch.from_numpy(G[pre + name + "_fc"]), torch.from_numpy(G[pre + name + "_f0"])
        close(t.sum((2, 3)), ref_fc, 2e-3, 2e-3 * float(ref_fc.abs().max()), name)
        close(t[0], ref_f0, 2e-3, 2e-3 * float(ref_f0.abs().max()), name + "_f0")      # the arg-max position receives the gradient
    bb = student.backbone
    for h, fc in ((1, bb.fc1), (2, bb.fc2)):
        rw, rb = torch.from_numpy(G[pre + "g_fc%d_w" % h]), torch.from_numpy(G[pre + "g_fc%d_b" % h])
        close(fc.weight.grad.sum(1), rw, 2e-3, 2e-3 * float(rw.abs().max()), "fc%d.weight" % h)
        close(fc.bias.grad, rb, 2e-3, 2e-3 * float(rb.abs().max()), "fc%d.bias" % h)


@pytest.mark.parametrize("proj", ["gemm", "conv"])
@pytest.mark.parametrize("case", [0, 1, 2])
def test_trx_golden(dev, golden_dir, case, proj):
    """TRX_2fcsup / TRX_2fcsup_fixed vs the fixtures produced by the reference's own module.  proj = conv: the four projections and
    their input gradient as 1x1 convolutions on the bf16-plane patch kernel (ops.TRX_PROJ_ON_CONV, opt-in) - same fixtures"""
    import os
    from litemkd_amd import ops
    from litemkd_amd.model import classifiers as C
    from oracle.gen_golden import trx_case_inputs
    was = ops.TRX_PROJ_ON_CONV
    ops.TRX_PROJ_ON_CONV = proj == "conv"
    try:
        _trx_golden(dev, golden_dir, case)
    finally:
        ops.TRX_PROJ_ON_CONV = was


def _trx_golden(dev, golden_dir, case):
    import os
    from litemkd_amd.model import classifiers as C
    from oracle.gen_golden import trx_case_inputs
    G = np.load(os.path.join(golden_dir, "trx.npz"))
    pre = "c%d_" % case
    ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
    p, sup1, qry1, sup2, qry2, lab = trx_case_inputs(int(G[pre + "seed"]), ns, nq, bool(G[pre + "shuffle"]))
    args = _args(dev, shot=ns // 5)
    clf = C.TRX_2fcsup(args)
    sd = clf.state_dict()
    for k, v in p.items():
        sd["transformers." + k].copy_(v)
    clf = clf.to(dev)
    t = [x.to(dev).requires_grad_() for x in (sup1, qry1, sup2, qry2)]
    r = clf({"context_features_1": t[0], "context_features_2": t[2]}, lab.to(dev),
            {"target_features_1": t[1], "target_features_2": t[3]})["logits"]
    for k in ("kl", "ce", "sup"):
        close(r[k], torch.from_numpy(G[pre + k]), 1e-4, 2e-2, "trx " + k)
    w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5).to(dev)
    w_sup = torch.linspace(1, -1, 20).reshape(5, 4).to(dev)
    ((r["kl"] * w).sum() * 1e-2 + (r["ce"] * w.flip(0)).sum() * 1e-2 + (r["sup"] * w_sup).sum() * 1e-3).backward()
    for name, x in zip(("g_sup1", "g_qry1", "g_sup2", "g_qry2"), t):
        ref = torch.from_numpy(G[pre + name])
        close(_gsum(x.grad), ref, 2e-3, 2e-3 * float(ref.abs().max()), name)
    close(t[0].grad[0, :, :128], torch.from_numpy(G[pre + "g_sup1_row0"]), 2e-3, 1e-5, "g_sup1 row0")
    tr = clf.transformers
    for name, gt in (("g_kb", tr.k_linear.bias.grad), ("g_vb", tr.v_linear.bias.grad), ("g_nkw", tr.norm_k.weight.grad),
                     ("g_nkb", tr.norm_k.bias.grad), ("g_kw_sum", tr.k_linear.weight.grad.sum(1)),
                     ("g_vw_sum", tr.v_linear.weight.grad.sum(1))):
        ref = torch.from_numpy(G[pre + name])
        # (the v-bias gradient is exactly 0 in exact arithmetic: softmax rows sum to 1 — compare against an absolute floor)
        close(gt, ref, 5e-3, 5e-3 * max(float(ref.abs().max()), 1e-4), name)
    fx = C.TRX_2fcsup_fixed(args)
    fsd = fx.state_dict()
    for k, v in p.items():
        fsd["transformers." + k].copy_(v)
    fx = fx.to(dev)
    rf = fx(sup1.to(dev), lab.to(dev), qry1.to(dev))["logits"]
    close(rf["kl"], torch.from_numpy(G[pre + "fixed_kl"]), 1e-4, 2e-2, "fixed kl")
    close(rf["sup"], torch.from_numpy(G[pre + "fixed_sup"]), 1e-4, 2e-2, "fixed sup")


@pytest.mark.parametrize("case", [0, 1])
def test_trx_sup_golden(dev, golden_dir, case):
    """TRX_sup / TRX_sup_fixed (TRX_sup.py) and Distiller.support_sim vs fixtures produced by the reference's own modules."""
    import os
    from litemkd_amd.model import classifiers as C
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import DEFAULT_CFG
    from oracle.gen_golden import trx_case_inputs
    G = np.load(os.path.join(golden_dir, "trx_sup.npz"))
    pre = "c%d_" % case
    ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
    p, sup, qry, sup_t, qry_t, lab = trx_case_inputs(int(G[pre + "seed"]), ns, nq, bool(G[pre + "shuffle"]))
    args = _args(dev, shot=ns // 5)
    clf, fx = C.TRX_sup(args), C.TRX_sup_fixed(args)
    sd, fsd = clf.state_dict(), fx.state_dict()
    for k, v in p.items():
        sd["transformers." + k].copy_(v)
        fsd["transformers." + k].copy_(0.9 * v)
    clf, fx = clf.to(dev), fx.to(dev)
    s, q = sup.to(dev).requires_grad_(), qry.to(dev).requires_grad_()
    r = clf(s, lab.to(dev), q)["logits"]
    rt = fx(sup_t.to(dev), lab.to(dev), qry_t.to(dev))["logits"]
    close(r["query"], torch.from_numpy(G[pre + "query"]), 1e-4, 2e-2, "query logits")
    close(r["support_set"], torch.from_numpy(G[pre + "support_set"]), 1e-4, 1e-5, "prototype similarities")
    close(rt["query"], torch.from_numpy(G[pre + "fixed_query"]), 1e-4, 2e-2, "fixed query logits")
    close(rt["support_set"], torch.from_numpy(G[pre + "fixed_support_set"]), 1e-4, 1e-5, "fixed similarities")
    labels = (torch.arange(nq) % 5).to(dev)
    res = Distiller("support_sim", dict(DEFAULT_CFG), dev).support_sim(r, rt, labels)
    for k in ("loss", "hard_loss", "soft_support_loss", "soft_query_loss"):
        close(res[k], torch.from_numpy(G[pre + k]), 1e-4, 1e-5, k)
    w = torch.linspace(-1, 1, nq * 25).reshape(nq, 5, 5).to(dev)
    (res["loss"] + (r["support_set"] * w).sum() * 1e-2).backward()
    for name, x in (("g_sup", s), ("g_qry", q)):
        ref = torch.from_numpy(G[pre + name])
        close(_gsum(x.grad), ref, 2e-3, 2e-3 * float(ref.abs().max()), name)
    tr = clf.transformers
    for name, gt in (("g_nkw", tr.norm_k.weight.grad), ("g_kw_sum", tr.k_linear.weight.grad.sum(1)), ("g_vw_sum", tr.v_linear.weight.grad.sum(1))):
        ref = torch.from_numpy(G[pre + name])
        close(gt, ref, 5e-3, 5e-3 * max(float(ref.abs().max()), 1e-4), name)


@pytest.mark.parametrize("way,shot", [(2, 3), (5, 5), (11, 2), (12, 2), (16, 1), (16, 3)])
def test_supportdk_all_ways(dev, way, shot):
    """SupportDK (TRX_2fcsup.py:162-189) for every supported way (<= LMKD_MAX_SEG = 16: up to 120 class pairs), forward and
    backward against the oracle.  way >= 12 has more than 64 pairs: the finishing kernel must visit all of them (a round-2
    regression left the pairs from 64 up unwritten; the output buffer is poisoned here so a missed pair shows)."""
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(100 * way + shot)
    sup = (torch.randn(way * shot, 8, 2048, generator=g) * 0.5).requires_grad_()
    gout = torch.randn(way, way - 1, generator=g)
    ref = O.support_dk(sup, way, shot)
    ref.backward(gout)
    torch.full((way, way - 1), float("nan"), device=dev)      # same-size block into the allocator's cache: a missed element reads NaN
    torch.cuda.synchronize()
    sd = sup.detach().to(dev).requires_grad_()
    out = ops.SupportDKFn.apply(sd, way, shot)
    out.backward(gout.to(dev))
    assert torch.isfinite(out).all()
    close(out, ref, 1e-4, 1e-3, "supportdk way %d" % way)
    close(sd.grad, sup.grad, 1e-3, 1e-5 * float(sup.grad.abs().max()) + 1e-7, "supportdk backward way %d" % way)


def test_trx_ragged_classes(dev):
    """unequal shots per class + a missing class (column stays 0, like torch.zeros in the reference)"""
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(77)
    p = O.make_trx_params(g)
    lab = torch.tensor([0., 0., 0., 2., 2., 3., 3., 3., 3., 4.])
    sup = torch.randn(10, 8, 2048, generator=g) * 0.5
    qry = torch.randn(6, 8, 2048, generator=g) * 0.5
    ref = O.trx_logits(sup, lab, qry, p, way=5)
    plan = ops.ClassPlan(lab.to(dev), 5)
    pd = {k: v.to(dev) for k, v in p.items()}
    out = ops.trx_logits_nograd(sup.to(dev), qry.to(dev), plan, pd["k_linear.weight"], pd["k_linear.bias"], pd["v_linear.weight"],
                                pd["v_linear.bias"], pd["norm_k.weight"], pd["norm_k.bias"], pd["pe.pe"][0, :8].contiguous())
    close(out, ref, 1e-4, 2e-2, "ragged trx")
    assert float(out[:, 1].abs().max()) == 0.0


@pytest.mark.parametrize("case", [0, 1, 2])
def test_edist_golden(dev, golden_dir, case):
    import os
    from litemkd_amd.model import classifiers as C
    from oracle.gen_golden import feature_case
    G = np.load(os.path.join(golden_dir, "edist.npz"))
    pre = "c%d_" % case
    ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
    sup, lab, qry = feature_case(int(G[pre + "seed"]), ns, nq, 1.0, bool(G[pre + "shuffle"]))
    args = _args(dev, shot=ns // 5)
    clf = C.e_dist_1fc_sup(args)
    s, q = sup.to(dev).requires_grad_(), qry.to(dev).requires_grad_()
    r = clf(s, lab.to(dev), q)["logits"]
    close(r["kl"], torch.from_numpy(G[pre + "kl"]), 1e-5, 1e-4, "edist kl")
    close(r["sup"], torch.from_numpy(G[pre + "sup"]), 1e-5, 1e-2, "supportdk")
    w_kl = torch.linspace(-1, 1, nq * 5).reshape(nq, 5).to(dev)
    w_sup = torch.linspace(1, -1, 20).reshape(5, 4).to(dev)
    ((r["kl"] * w_kl).sum() + (r["sup"] * w_sup).sum() * 1e-3).backward()
    close(_gsum(s.grad), torch.from_numpy(G[pre + "g_sup_feat"]), 1e-3, 1e-4, "edist dsup")
    close(_gsum(q.grad), torch.from_numpy(G[pre + "g_qry_feat"]), 1e-3, 1e-4, "edist dqry")
    close(s.grad[0, :, :128], torch.from_numpy(G[pre + "g_sup_row0"]), 1e-3, 1e-6, "edist dsup row0")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_distill_golden(dev, golden_dir, case):
    import os
    from litemkd_amd import distillers as D
    from litemkd_amd import ops
    from litemkd_amd.options import DEFAULT_CFG
    G = np.load(os.path.join(golden_dir, "distill.npz"))
    pre = "c%d_" % case
    T = lambda k: torch.from_numpy(G[pre + k]).to(dev)          # noqa: E731
    s = {k: T("s_" + k).clone().requires_grad_() for k in ("kl", "ce", "sup")}
    t = {k: T("t_" + k) for k in ("kl", "sup")}
    labels = T("labels")
    dist = D.Distiller("fc_2_sup_dist", dict(DEFAULT_CFG), dev)
    r = dist.fc_2_sup_dist(s, t, labels)
    r["loss"].backward()
    close(r["loss"], T("loss"), 1e-5, 1e-5, "loss")
    close(r["soft_loss"], T("soft"), 1e-5, 1e-5, "soft")
    close(r["hard_loss"], T("hard"), 1e-5, 1e-5, "hard")
    for k in ("kl", "ce", "sup"):
        close(s[k].grad, T("g_" + k), 1e-4, 1e-7, "grad " + k)
    close(D.kd_loss(s["kl"].detach(), t["kl"], 4), T("kd"), 1e-5, 1e-5, "kd_loss")
    close(D.inter_class_relation(s["sup"].detach(), t["sup"]), T("icr"), 1e-5, 1e-5, "icr")
    s1 = T("s_kl").clone().requires_grad_()
    r = dist.KD(s1, t["kl"], labels)
    r["loss"].backward()
    close(r["loss"], T("KD_loss"), 1e-5, 1e-5, "KD")
    close(s1.grad, T("KD_g"), 1e-4, 1e-7, "KD grad")
    s2 = T("s_kl").clone().requires_grad_()
    r = dist.Dist_KD(s2, t["kl"], labels)
    r["loss"].backward()
    close(r["loss"], T("DistKD_loss"), 1e-5, 1e-5, "Dist_KD")
    close(s2.grad, T("DistKD_g"), 1e-4, 1e-7, "Dist_KD grad")
    acc, pred = ops.accuracy(s["kl"].detach(), s["ce"].detach(), labels)
    assert np.array_equal(pred.cpu().numpy(), G[pre + "argmax"])            # bit-exact class indices
    assert float(acc) == float(G[pre + "acc"])


@pytest.mark.parametrize("d,N", [(256, 7), (512, 10)])
def test_mfm_fusion(dev, d, N):
    """MFM fusion (ThreeTRXShiftLoopTime.extract_feature) vs the oracle restatement, reduced width (same code path;
    the full 2048-wide model has 541 M parameters).  PARITY UNPINNED by the reference (teacher/code/model.py does not
    import here) — the oracle restates nn.TransformerEncoderLayer's documented defaults."""
    import argparse
    from litemkd_amd.teacher import ThreeTRXShiftLoopTime
    from oracle import ref_cpu as O
    args = argparse.Namespace(seq_len=8, trans_num=2, shirt_num=1, trans_linear_in_dim=d)
    torch.manual_seed(4)
    m = ThreeTRXShiftLoopTime(args, in_channels=d).eval()
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    rgb, depth, flow = (torch.randn(N, 8, d, generator=g).abs() for _ in range(3))
    ref = O.mfm_extract_feature(rgb, depth, flow, p, 1, 2)
    # cross-check the oracle itself against torch's own TransformerEncoder modules on the CPU
    with torch.no_grad():
        tf = m.three_fusion
        h = torch.cat([F.layer_norm(x + getattr(tf, "positionEncoding%d" % (i + 1)).position_embeddings.weight[:8],
                                    (d,), getattr(tf, "positionEncoding%d" % (i + 1)).LayerNorm.weight,
                                    getattr(tf, "positionEncoding%d" % (i + 1)).LayerNorm.bias) for i, x in enumerate((rgb, depth, flow))], -1)
        t_ref = tf.f1(tf.transformer_encoder(h))
        close(O.mfm_three_fusion(rgb, depth, flow, p), t_ref, 1e-4, 1e-4, "oracle vs nn.TransformerEncoder")
    m = m.to(dev)
    out = m.extract_feature({"rgb": rgb.to(dev), "depth": depth.to(dev), "flow": flow.to(dev)})
    close(out, ref, 1e-3, 1e-3, "mfm fused feature")


def test_mfm_fusion_full_width_golden(dev, golden_dir):
    """The HIP MFM fusion at the reference's fixed width 2048 (541 M parameters) against the outputs of the reference's own
    ThreeTRXShiftLoopTime.extract_feature (teacher/code/model.py:1648-1664; tests/golden/mfm.npz, weights from
    oracle.make_mfm_params): fused feature, both cases (shirt_num 1 / 2), and the partial encoders."""
    import argparse
    from litemkd_amd.teacher import ThreeTRXShiftLoopTime
    from oracle import ref_cpu as O
    G = np.load(os.path.join(golden_dir, "mfm.npz"))
    args = argparse.Namespace(seq_len=8, trans_num=2, shirt_num=1, trans_linear_in_dim=2048)
    with torch.device(dev):
        m = ThreeTRXShiftLoopTime(args).eval()
    p = O.make_mfm_params(int(G["weight_seed"]))
    m.load_state_dict(p, strict=True)
    del p
    for case in (0, 1):
        pre = "c%d_" % case
        n, shirt = int(G[pre + "n"]), int(G[pre + "shirt"])
        rgb, depth, flow = (t.to(dev) for t in O.make_mfm_inputs(int(G[pre + "seed"]), n))
        args.shirt_num = shirt
        out = m.extract_feature({"rgb": rgb, "depth": depth, "flow": flow})
        # fp32 GEMMs with K up to 6144 and a different summation order than the CPU: 2e-4 of the feature scale (~1)
        close(out, torch.from_numpy(G[pre + "out"]), 2e-4, 2e-4, "mfm fused feature, case %d" % case)
        close(m.three_fusion.extract_feature(rgb, depth, flow), torch.from_numpy(G[pre + "three"]), 2e-4, 2e-4, "three_fusion")
        close(m.fusion.extract_feature(rgb, torch.roll(depth, -shirt, 1).contiguous()), torch.from_numpy(G[pre + "two_depth"]),
              2e-4, 2e-4, "fusion(rgb, depth)")


def test_kl_feature_golden(dev, golden_dir):
    """Distiller.KL_feature (lmkd_d2m_loss + lmkd_mse_loss) vs the reference's own method: loss terms, logits gradient and the
    feature gradient 2 w (s - t) / n"""
    from litemkd_amd.distillers import Distiller
    from oracle import ref_cpu as O
    from oracle.gen_golden import kl_feature_inputs, gsum
    G = np.load(os.path.join(golden_dir, "kl_feature.npz"))
    for case in (0, 1):
        pre = "c%d_" % case
        s, t, labels = kl_feature_inputs(int(G[pre + "seed"]), int(G[pre + "nq"]), int(G[pre + "nv"]))
        s = {k: v.to(dev).requires_grad_() for k, v in s.items()}
        t = {k: v.to(dev) for k, v in t.items()}
        r = Distiller("KL_feature", dict(O.DEFAULT_CFG), dev).KL_feature(s, t, labels.to(dev))
        r["loss"].backward()
        close(r["loss"], torch.from_numpy(G[pre + "loss"]), 1e-5, 1e-5, "KL_feature loss")
        close(r["feature_loss"], torch.from_numpy(G[pre + "feature_loss"]), 1e-5, 1e-6, "feature loss")
        close(r["hard_loss"], torch.from_numpy(G[pre + "hard"]), 1e-5, 1e-6, "hard loss")
        close(r["soft_loss"], torch.from_numpy(G[pre + "soft"]), 1e-5, 1e-5, "soft loss")
        close(s["logits"].grad, torch.from_numpy(G[pre + "g_logits"]), 1e-4, 1e-7, "logits grad")
        close(gsum(s["feature"].grad.cpu()), torch.from_numpy(G[pre + "g_feature"]), 1e-4, 1e-8, "feature grad (chunk sums)")
        close(s["feature"].grad[0, :, :128], torch.from_numpy(G[pre + "g_feature_row0"]), 1e-5, 1e-10, "feature grad row 0")


def test_all_distiller_methods_golden(dev, golden_dir):
    """every logits-only Distiller method against the fixtures produced by the reference's distillers.py"""
    import os
    from litemkd_amd import distillers as D
    from litemkd_amd.options import DEFAULT_CFG
    from oracle import ref_cpu as O
    G = np.load(os.path.join(golden_dir, "distill_methods.npz"))
    for i, name in enumerate(sorted(O.DISTILL_SIGNATURES)):
        s, t, labels = O.distill_inputs(name, 500 + i)
        mv = lambda x: x.to(dev) if torch.is_tensor(x) else {k: v.to(dev) for k, v in x.items()}      # noqa: E731
        s, t, labels = mv(s), mv(t), labels.to(dev)
        leaves = {"": s} if torch.is_tensor(s) else s
        for v in leaves.values():
            v.requires_grad_()
        r = getattr(D.Distiller(name, dict(DEFAULT_CFG), dev), name)(s, t, labels)
        r["loss"].backward()
        close(r["loss"], torch.from_numpy(G[name + "__loss"]), 1e-5, 1e-5, name)
        for k, v in leaves.items():
            ref = torch.from_numpy(G[name + ("__g" if k == "" else "__g_" + k)])
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            close(got, ref, 1e-4, 1e-7, name + " grad " + k)


def test_resnet50_trunk_forward_backward(dev):
    """ResNet-50 trunk (Bottleneck blocks, BatchNorm over up to 2048 channels) fwd + bwd vs the oracle, 6 frames 96x96"""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone.resnet import ResNet50Trunk
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(8)
    shapes = O.resnet50_trunk_param_shapes()
    n = sum(int(np.prod(s)) for k, s in shapes.items() if "running" not in k and "num_batches" not in k)
    assert n == 23508032                                  # torchvision resnet50 minus the 2048x1000 fc
    sd = O.init_trunk_from_shapes(shapes, g)
    trunk = ResNet50Trunk()
    trunk.load_state_dict(sd)
    trunk = trunk.to(dev).train()
    x = torch.rand(6, 3, 96, 96, generator=g)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    fm = O.resnet50_trunk(x, osd, True, True)
    feat = O.pooled_frame_features(fm)
    gfeat = rnd(*feat.shape, seed=9)
    feat.backward(gfeat)
    y = trunk(x.to(dev))
    close(nchw(y), fm, 2e-3, 2e-3 * float(fm.abs().max()), "resnet50 feature map")
    f = ops.PoolHeadFn.apply(y)
    f.backward(gfeat.to(dev))
    close(trunk.state_dict()["7.2.bn3.running_var"], osd["7.2.bn3.running_var"], 1e-3, 1e-4, "running_var")
    worst = 0.0
    for k, p in trunk.named_parameters():
        ref = osd[k].grad.double()
        err = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
        worst = max(worst, err)
        assert err < 5e-2, "grad %s: rel-L2 err %.3e" % (k, err)
    print("resnet50 worst grad rel-L2 err:", worst)


def test_frames_u8_transform(dev):
    """GPU crop + flip + ToTensor of uint8 frames == the reference's transform order (flip, then crop, then ToTensor:
    video_reader.py:92-112) and the stem accepts the NHWC4 result"""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone.resnet import ResNet18Trunk
    g = torch.Generator().manual_seed(3)
    F_, Hs, Ws, S = 16, 72, 80, 64
    u8 = torch.randint(0, 256, (F_, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    cy = torch.tensor([3, 8], dtype=torch.int32)
    cx = torch.tensor([10, 0], dtype=torch.int32)
    fl = torch.tensor([1, 0], dtype=torch.int32)
    ref = torch.empty(F_, 3, S, S)
    for f in range(F_):
        v = f // 8
        img = u8[f].float() / 255.0                      # ToTensor commutes with crop / flip
        img = img[int(cy[v]):int(cy[v]) + S, int(cx[v]):int(cx[v]) + S]
        if int(fl[v]):
            img = img.flip(1)
        ref[f] = img.permute(2, 0, 1)
    out = ops.frames_u8_to_nhwc4(u8.to(dev), cy.to(dev), cx.to(dev), fl.to(dev), S)
    assert torch.equal(out[..., :3].cpu(), ref.permute(0, 2, 3, 1)) and float(out[..., 3].abs().max()) == 0.0
    torch.manual_seed(1)
    trunk = ResNet18Trunk().to(dev).eval()
    with torch.no_grad():
        assert torch.equal(trunk(out), trunk(ref.to(dev)))


@pytest.fixture
def bf16_convs():
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("bf16")
    yield
    ops.reset_compute_dtypes()


@pytest.mark.parametrize("cfg", [(3, 3, 64, 64, 64, 7, 2, 3), (2, 64, 28, 28, 64, 3, 1, 1), (2, 64, 28, 28, 128, 3, 2, 1),
                                 (2, 128, 14, 14, 128, 3, 1, 1), (5, 256, 7, 7, 512, 3, 1, 1), (2, 64, 28, 28, 128, 1, 2, 0)])
def test_conv_bf16_mode(dev, bf16_convs, cfg):
    """BASELINE configs[2] arithmetic: operands rounded to bf16 (RNE), fp32 accumulation.  Oracle = the fp32 convolution of
    the bf16-rounded tensors, so only the summation order differs (same tolerance as the fp32 test)."""
    from litemkd_amd import ops
    N, Cin, H, W, Cout, K, s, p = cfg
    bf = lambda t: t.bfloat16().float()            # noqa: E731
    x = bf(rnd(N, Cin, H, W, seed=10)).requires_grad_()
    w = bf(rnd(Cout, Cin, K, K, seed=11) * math.sqrt(2.0 / (Cout * K * K))).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    gy = bf(rnd(*y.shape, seed=12))
    y.backward(gy)
    Cs = 4 if Cin == 3 else Cin
    xd = torch.zeros(N, H, W, Cs)
    xd[..., :Cin] = nhwc(x.detach())
    xd, wdv = xd.to(dev), w.detach().to(dev)
    yd, part = ops.conv_fwd(xd, ops._pack_weights(wdv, Cs, 0), Cout, K, K, s, p, True)
    close(nchw(yd), y, 1e-4, 2e-5 * math.sqrt(Cin * K * K), "bf16 conv fwd")
    close(part.double().sum(0).cpu()[:, 0], y.detach().double().sum((0, 2, 3)), 1e-4, 1e-2, "bn sum")
    gyd = nhwc(gy).to(dev)
    dw = ops.conv_bwd_weight(xd, gyd, tuple(w.shape), s, p)
    close(dw, w.grad, 1e-3, 2e-5 * math.sqrt(N * y.shape[2] * y.shape[3]) * 3, "bf16 conv wgrad")
    if Cin != 3:
        dx = ops.conv_bwd_data(gyd, ops._pack_weights(wdv, Cin, 1), (N, H, W, Cin), Cout, K, K, s, p)
        close(nchw(dx), x.grad, 1e-4, 2e-5 * math.sqrt(Cout * K * K), "bf16 conv dgrad")
    # and the mode really rounds: an operand that is NOT bf16-representable gives a different result than fp32 mode
    x2 = xd + 1e-3
    y_b, _ = ops.conv_fwd(x2, ops._pack_weights(wdv, Cs, 0), Cout, K, K, s, p, False)
    ops.set_conv_compute_dtype("fp32")
    y_f, _ = ops.conv_fwd(x2, ops._pack_weights(wdv, Cs, 0), Cout, K, K, s, p, False)
    ops.set_conv_compute_dtype("bf16")
    assert float((y_b - y_f).abs().max()) > 0


def test_episode_bf16_close_to_fp32(dev, bf16_convs):
    """whole training episode with bf16 convolutions vs the same episode in fp32 on the GPU: features within 3 % of their
    max (bf16 has an 8-bit mantissa), loss within 2 %, finite gradients.  Tolerance of BASELINE configs[2]."""
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    args = default_args(shot=1, query_per_class=1, img_size=96, trans_dropout=0.0, device=dev)
    torch.manual_seed(1)
    student, teacher = Student(args).to(dev), Teacher(args).to(dev)
    ep = O.make_episode(901, 5, 1, 1, img=96)
    labels = ep["target_labels"].long().to(dev)

    def run():
        student.zero_grad()
        out = student(ep["support_set"].to(dev), ep["support_labels"].to(dev), ep["target_set"].to(dev))
        tl = teacher(ep["support_set_feature_teacher"].to(dev), ep["support_labels"].to(dev), ep["target_set_feature_teacher"].to(dev))["logits"]
        loss = Distiller("fc_2_sup_dist", args.cfg, dev).fc_2_sup_dist(out["logits"], tl, labels)["loss"]
        loss.backward()
        return out, loss.item(), [p.grad.clone() for p in student.parameters() if p.grad is not None]
    out_b, loss_b, g_b = run()
    ops.set_conv_compute_dtype("fp32")
    out_f, loss_f, g_f = run()
    ops.set_conv_compute_dtype("bf16")
    for k in ("context_features_1", "context_features_2"):
        a, b = out_b["context_features"][k], out_f["context_features"][k]
        assert float((a - b).abs().max()) < 3e-2 * float(b.abs().max()), k
        assert float((a - b).abs().max()) > 0
    assert abs(loss_b - loss_f) < 2e-2 * abs(loss_f)
    assert all(torch.isfinite(g).all() for g in g_b) and len(g_b) == len(g_f)


@pytest.fixture
def x3_convs():
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("fp32x3")
    yield
    ops.reset_compute_dtypes()


@pytest.mark.parametrize("cfg", [(3, 3, 64, 64, 64, 7, 2, 3), (2, 64, 28, 28, 64, 3, 1, 1), (2, 64, 28, 28, 128, 3, 2, 1),
                                 (2, 64, 27, 29, 128, 3, 2, 1), (9, 128, 14, 14, 128, 3, 1, 1), (5, 256, 7, 7, 512, 3, 1, 1),
                                 (2, 64, 28, 28, 128, 1, 2, 0), (40, 128, 28, 28, 256, 3, 1, 1)])
@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9"])
def test_conv_x3_mode(dev, cfg, mode):
    """fp32 convolution on the bf16 matrix pipe (exact 3-way bf16 split of both operands, 6 or 9 products, fp32 accumulate,
    csrc/conv_x3.h).  Same tolerance as the native fp32 test, and measured against an fp64 convolution its relative-L2 error
    stays within 2x of the native fp32 MFMA kernel's (measured 0.7x - 1.4x; both are a few 1e-7): fp32-class arithmetic,
    not reduced precision."""
    from litemkd_amd import ops
    N, Cin, H, W, Cout, K, s, p = cfg
    x = rnd(N, Cin, H, W, seed=20).relu().double().requires_grad_()
    w = (rnd(Cout, Cin, K, K, seed=21) * math.sqrt(2.0 / (Cout * K * K))).double().requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    gy = rnd(*y.shape, seed=22).double()
    y.backward(gy)
    Cs = 4 if Cin == 3 else Cin
    xd = torch.zeros(N, H, W, Cs)
    xd[..., :Cin] = nhwc(x.detach().float())
    xd, wdv, gyd = xd.to(dev), w.detach().float().to(dev), nhwc(gy.float()).to(dev)
    res = {}
    try:
        for m in ("fp32", mode):
            ops.set_conv_compute_dtype(m)
            yd, part = ops.conv_fwd(xd, ops._pack_weights(wdv, Cs, 0), Cout, K, K, s, p, True)
            dx = None if Cin == 3 else ops.conv_bwd_data(gyd, ops._pack_weights(wdv, Cin, 1), (N, H, W, Cin), Cout, K, K, s, p)
            res[m] = (yd, part, dx, ops.conv_bwd_weight(xd, gyd, tuple(w.shape), s, p))
    finally:
        ops.reset_compute_dtypes()
    yd, part, dx, dw = res[mode]
    close(nchw(yd), y.float(), 1e-4, 2e-5 * math.sqrt(Cin * K * K), "x3 conv fwd")
    close(dw, w.grad.float(), 1e-3, 2e-5 * math.sqrt(N * y.shape[2] * y.shape[3]) * 3, "x3 conv wgrad")
    close(part.double().sum(0).cpu()[:, 0], y.detach().sum((0, 2, 3)), 1e-4, 1e-2, "bn sum")
    close(part.double().sum(0).cpu()[:, 1], (y.detach() ** 2).sum((0, 2, 3)), 1e-4, 1e-2, "bn sum of squares")
    err = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())        # noqa: E731
    e_nat, e_x3 = err(nchw(res["fp32"][0]), y.detach()), err(nchw(yd), y.detach())
    assert e_x3 <= 2.0 * e_nat + 1e-8, (e_x3, e_nat)
    # weight gradient: the window kernel (3x3 / stride 1) sums a slab's pixels in one accumulator chain, the native kernel in other
    # portions - bound the error by the native kernel's (2x) or by torch-CPU fp32's (3x, the criterion of tests/_anchor.py)
    w32 = w.detach().float().requires_grad_()
    F.conv2d(x.detach().float(), w32, None, s, p).backward(gy.float())
    e_nat, e_x3, e_cpu = err(res["fp32"][3], w.grad), err(dw, w.grad), err(w32.grad, w.grad)
    assert e_x3 <= max(2.0 * e_nat, 3.0 * e_cpu) + 1e-8, ("wgrad", e_x3, e_nat, e_cpu)
    if dx is not None:
        close(nchw(dx), x.grad.float(), 1e-4, 2e-5 * math.sqrt(Cout * K * K), "x3 conv dgrad")
        e_nat, e_x3 = err(nchw(res["fp32"][2]), x.grad), err(nchw(dx), x.grad)
        assert e_x3 <= 2.0 * e_nat + 1e-8, (e_x3, e_nat)


def test_episode_x3_matches_fp32(dev, x3_convs):
    """whole training episode with the 3xbf16 convolutions vs the same episode with the fp32 MFMA convolutions: same loss
    to 1e-5 relative, features to 1e-4 of their max, gradients to 2e-2 relative L2 (ReLU-mask flips, as between any two fp32
    implementations; see test_block_isolated)."""
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    args = default_args(shot=1, query_per_class=1, img_size=96, trans_dropout=0.0, device=dev)
    torch.manual_seed(1)
    student, teacher = Student(args).to(dev), Teacher(args).to(dev)
    ep = O.make_episode(902, 5, 1, 1, img=96)
    labels = ep["target_labels"].long().to(dev)

    def run():
        student.zero_grad()
        out = student(ep["support_set"].to(dev), ep["support_labels"].to(dev), ep["target_set"].to(dev))
        tl = teacher(ep["support_set_feature_teacher"].to(dev), ep["support_labels"].to(dev), ep["target_set_feature_teacher"].to(dev))["logits"]
        loss = Distiller("fc_2_sup_dist", args.cfg, dev).fc_2_sup_dist(out["logits"], tl, labels)["loss"]
        loss.backward()
        return out, loss.item(), {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
    out_x, loss_x, g_x = run()
    ops.set_conv_compute_dtype("fp32")
    out_f, loss_f, g_f = run()
    ops.set_conv_compute_dtype("fp32x3")
    for k in ("context_features_1", "context_features_2"):
        a, b = out_x["context_features"][k], out_f["context_features"][k]
        assert float((a - b).abs().max()) < 1e-4 * float(b.abs().max()), k
    assert abs(loss_x - loss_f) < 1e-5 * abs(loss_f)
    num = sum(float((g_x[n] - g_f[n]).double().pow(2).sum()) for n in g_f)
    den = sum(float(g_f[n].double().pow(2).sum()) for n in g_f)
    assert math.sqrt(num / den) < 2e-2


@pytest.mark.parametrize("mode", ["fp32x3", "fp32", "fp32x3_9", "bf16", "fp32h2"])
@pytest.mark.parametrize("arch,frames,img", [("resnet18", 6, 64), ("resnet50", 6, 64), ("resnet18", 40, 224)])
def test_eval_bn_fused_in_conv_epilogue(dev, arch, frames, img, mode):
    """inference path (model.eval(), trainwandb.py:366): BatchNorm + residual + ReLU run in the convolution epilogue
    (lmkd_conv2d_fwd_bn) in EVERY arithmetic mode - conv_gemm_kernel's STATS == 2 epilogue in the native fp32 mode, x3_epilogue<EP> of
    the patch / gather plane kernels in the others (round 3).  Must be BIT-identical to the two-pass form (conv, then bn_apply) and
    match the oracle trunk.  The 40-frame 224^2 case runs the tile instances of a full-size launch (128x128 patch tiles, XCD orders)."""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone.resnet import ResNet18Trunk, ResNet50Trunk
    from oracle import ref_cpu as O
    ops.set_conv_compute_dtype(mode)
    torch.manual_seed(3)
    trunk = (ResNet18Trunk() if arch == "resnet18" else ResNet50Trunk()).to(dev)
    with torch.no_grad():                      # non-trivial running statistics
        for n, b in trunk.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn_like(b) * 0.1)
            elif n.endswith("running_var"):
                b.copy_(torch.rand_like(b) + 0.5)
    trunk.eval()
    x = torch.rand(frames, 3, img, img)
    import litemkd_amd
    h2_0 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    with torch.no_grad():
        assert ops._eval_fused()
        y_fused = trunk(x.to(dev))
        if mode == "fp32h2":
            # round 5: inference in the headline arithmetic - the fused epilogue records max |y| itself (the BatchNorm + ReLU are applied
            # there), so every same-size / stride-2 3x3 and same-size 1x1 launch of the eval forward runs two planes: ResNet-18 the stem + 16,
            # ResNet-50 the stem + 16 3x3 + 33 1x1 (the stride-2 1x1 downsample convolutions stay on the gather kernel's three planes)
            assert litemkd_amd.lib().value("lmkd_conv_h2_launches") - h2_0 == (17 if arch == "resnet18" else 50)
        ops.FUSE_EVAL_BN = False
        try:
            y_two = trunk(x.to(dev))
        finally:
            ops.FUSE_EVAL_BN = True
    if mode == "fp32h2":
        # the fused launches of the 64-channel layers run the 16x16x32 kernel, the two-pass form's convolutions the 32x32x16 one
        # (lmkd_conv_set_patch16: each the faster for its epilogue): the same arithmetic in another summation order - equal to fp32 rounding
        assert float((y_fused - y_two).abs().max()) <= 1e-5 * float(y_two.abs().max())
    else:
        assert torch.equal(y_fused, y_two)
    sd = {k: v.detach().cpu() for k, v in trunk.state_dict().items()}
    O.CONV_BF16 = mode == "bf16"
    try:
        ref = (O.resnet18_trunk if arch == "resnet18" else O.resnet50_trunk)(x, sd, training=False)
    finally:
        O.CONV_BF16 = False
        ops.reset_compute_dtypes()
    # bf16: the same operand rounding on both sides, but an activation within accumulation noise of a bf16 rounding boundary rounds
    # the other way (0.4 % of that operand) and the difference travels through the remaining layers: percent-level on single elements
    tol = 2e-2 if mode == "bf16" else 1e-4
    close(nchw(y_fused), ref, 5 * tol, tol * float(ref.abs().max()), "eval trunk vs oracle")


def test_resize_frames_matches_pillow(dev, golden_dir):
    """GPU Resize of the frame transform vs fixtures produced by the reference's functional.resize_clip with PIL (bit exact),
    a batch against the numpy restatement, and the whole transform chain Resize(256) -> crop/flip -> ToTensor into the stem layout."""
    import os
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    G = np.load(os.path.join(golden_dir, "resize.npz"))
    for c in range(4):
        img, ref, size = G["c%d_in" % c], G["c%d_out" % c], int(G["c%d_size" % c])
        out = ops.resize_frames_u8(torch.from_numpy(img)[None].to(dev), size)
        assert out.shape[1:] == ref.shape and np.array_equal(out[0].cpu().numpy(), ref), c
    rng = np.random.default_rng(9)
    batch = rng.integers(0, 256, (16, 120, 160, 3), dtype=np.uint8)
    out = ops.resize_frames_u8(torch.from_numpy(batch).to(dev), 128)
    assert out.shape == (16, 128, 170, 3)
    assert np.array_equal(out.cpu().numpy(), O.pil_resize_bilinear_u8(batch, 170, 128))
    # chain: Resize -> per-video crop + flip -> ToTensor (video_reader.py:96-112), 2 videos x 8 frames
    cy = torch.tensor([3, 10], dtype=torch.int32)
    cx = torch.tensor([40, 0], dtype=torch.int32)
    fl = torch.tensor([1, 0], dtype=torch.int32)
    x = ops.frames_u8_to_nhwc4(out, cy.to(dev), cx.to(dev), fl.to(dev), 112)
    r = torch.from_numpy(O.pil_resize_bilinear_u8(batch, 170, 128))
    for f in range(16):
        v = f // 8
        ref = r[f, int(cy[v]):int(cy[v]) + 112, int(cx[v]):int(cx[v]) + 112]
        if int(fl[v]):
            ref = ref.flip(1)
        assert torch.equal(x[f, :, :, :3].cpu(), ref.float() / 255.0)


def test_gpu_frame_transform_matches_reference_chain(dev):
    """GpuFrameTransform vs the reference's per-video chain Resize(256) -> RandomHorizontalFlip -> RandomCrop(224) -> ToTensor
    (video_reader.py:92-112,377-385) restated on the host with the same `random` stream; mixed source resolutions; test mode
    (CenterCrop) as well.  Bit exact."""
    import random
    from litemkd_amd.video_transform import GpuFrameTransform
    from oracle import ref_cpu as O
    rng = np.random.default_rng(12)
    vids = [rng.integers(0, 256, (8, h, w, 3), dtype=np.uint8) for (h, w) in [(240, 320), (240, 320), (288, 352), (240, 320), (256, 300)]]
    tf = GpuFrameTransform(224, dev)

    def host_chain(train):
        outs = []
        for v in vids:
            oh, ow = O.resize_short_side(v.shape[1], v.shape[2], 256)
            r = O.pil_resize_bilinear_u8(v, ow, oh)
            if train:
                flip = random.random() < 0.5
                x1 = random.randint(0, ow - 224)
                y1 = random.randint(0, oh - 224)
            else:
                flip, x1, y1 = False, int(round((ow - 224) / 2.)), int(round((oh - 224) / 2.))
            if flip:
                r = r[:, :, ::-1]
            c = np.ascontiguousarray(r[:, y1:y1 + 224, x1:x1 + 224])        # the reference flips first, then crops
            outs.append(torch.from_numpy(c).float() / 255.0)
        return torch.cat(outs, 0)
    for train in (True, False):
        random.seed(77)
        ref = host_chain(train)
        random.seed(77)
        out = tf([torch.from_numpy(v) for v in vids], train=train)
        assert out.shape == (40, 224, 224, 4)
        assert torch.equal(out[..., :3].cpu(), ref) and float(out[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("shape,resize,S", [((240, 320), 256, 224), ((288, 352), 256, 224), ((256, 300), 256, 224), ((120, 160), 128, 112),
                                            ((300, 200), 128, 112), ((64, 64), (80, 72), 64)])
def test_fused_frame_transform_equals_two_pass_and_pillow(dev, shape, resize, S):
    """lmkd_frames_resize_crop_nhwc4 (one launch per episode: Resize -> crop -> flip -> ToTensor, each thread evaluating Pillow's two
    passes for its own output pixel) against the numpy restatement of Pillow's resampler (oracle, pinned by tests/golden/resize.npz) and
    against the two-pass path (lmkd_resize_pass_u8 x 2 + lmkd_frames_u8_to_nhwc4): bit exact, flipped and unflipped videos, up- and
    down-scaling, crops at the frame's edges (video_reader.py:92-112,377-385)."""
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    rng = np.random.default_rng(21)
    H, W = shape
    L, nv = 4, 5
    frames = rng.integers(0, 256, (L * nv, H, W, 3), dtype=np.uint8)
    oh, ow = ops._resized_shape(H, W, resize)
    cy = torch.tensor([0, oh - S, (oh - S) // 2, 1 % (oh - S + 1), oh - S], dtype=torch.int32)
    cx = torch.tensor([0, ow - S, (ow - S) // 3, ow - S, 0], dtype=torch.int32)
    fl = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
    fr = torch.from_numpy(frames).to(dev)
    out = ops.frames_resize_crop_nhwc4(fr, resize, cy.to(dev), cx.to(dev), fl.to(dev), S, frames_per_video=L)
    two = ops.frames_u8_to_nhwc4(ops.resize_frames_u8(fr, resize), cy.to(dev), cx.to(dev), fl.to(dev), S, frames_per_video=L)
    assert out.shape == (L * nv, S, S, 4) and torch.equal(out, two)
    r = torch.from_numpy(O.pil_resize_bilinear_u8(frames, ow, oh))
    for f in range(L * nv):
        v = f // L
        ref = r[f, int(cy[v]):int(cy[v]) + S, int(cx[v]):int(cx[v]) + S]
        if int(fl[v]):
            ref = ref.flip(1)
        assert torch.equal(out[f, :, :, :3].cpu(), ref.float() / 255.0), f
    assert float(out[..., 3].abs().max()) == 0.0


@pytest.fixture
def bf16_act():
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("bf16")
    ops.set_activation_dtype("bf16")
    yield
    ops.set_activation_dtype("fp32")
    ops.reset_compute_dtypes()


@pytest.mark.parametrize("cfg", [(5, 64, 14, 14, 64, 3, 1, 1), (3, 128, 9, 11, 256, 3, 2, 1), (4, 256, 7, 7, 64, 1, 2, 0), (40, 128, 28, 28, 128, 3, 1, 1)])
def test_bf16_activation_kernels_equal_rounded_fp32_kernels(dev, cfg):
    """BASELINE configs[2] with bf16 tensors in HBM (lmkd_set_activation_dtype(1)): every kernel instance that reads / writes bf16
    activations must give exactly the ROUNDED result of the fp32-activation instance on the same (bf16-representable) inputs -
    the arithmetic in between is the same fp32 arithmetic; statistics are those of the stored (rounded) convolution output."""
    from litemkd_amd import ops
    N, C, H, W, Cout, K, s, p = cfg
    r = lambda t: t.to(torch.bfloat16)                                        # noqa: E731
    x32 = r(rnd(N, H, W, C, seed=60) * 1.5).float().to(dev)
    w = (rnd(Cout, C, K, K, seed=61) * math.sqrt(2.0 / (Cout * K * K))).to(dev)
    ops.set_conv_compute_dtype("bf16")
    try:
        wp, wd = ops._pack_weights(w, C, 0), ops._pack_weights(w, C, 1)
        y32, p32 = ops.conv_fwd(x32, wp, Cout, K, K, s, p, True)
        dy32 = r(rnd(*y32.shape, seed=62)).float().to(dev)
        dx32 = ops.conv_bwd_data(dy32, wd, (N, H, W, C), Cout, K, K, s, p)
        res32 = r(rnd(N, H, W, C, seed=63)).float().to(dev)
        acc32 = res32.clone()
        ops.conv_bwd_data(dy32, wd, (N, H, W, C), Cout, K, K, s, p, out=acc32, accumulate=True)
        # weight gradient: by either kernel family (rolling window for 3x3 / stride 1, im2col gather), compared family by family
        import ctypes
        import litemkd_amd
        L = litemkd_amd.lib()

        def wgrad_both(xx, dd):
            out = {}
            for win in (0, 1):
                L.call("lmkd_conv_set_wgrad_window", win)
                info = (ctypes.c_int * 5)()
                L.call("lmkd_conv2d_plan", 2, N, H, W, C, C, Cout, K, K, s, p, info)
                out[(info[0], info[4])] = ops.conv_bwd_weight(xx, dd, (Cout, C, K, K), s, p)
            L.call("lmkd_conv_set_wgrad_window", 1)
            return out
        dw32 = wgrad_both(x32, dy32)
        # BatchNorm / pooling on a [N,H,W,C] activation
        gamma, beta = (1 + 0.2 * rnd(C, seed=64)).to(dev), (0.3 * rnd(C, seed=65)).to(dev)
        flat = x32.reshape(-1, C)
        part = torch.stack([flat.sum(0, keepdim=True), (flat ** 2).sum(0, keepdim=True)], -1).contiguous()
        st = ops.bn_stats_train(part, flat.shape[0], gamma, beta, None, None)
        a32, bits32 = ops.bn_apply(x32, st, True, res32, want_bits=True)
        g32 = r(rnd(N, H, W, C, seed=66)).float().to(dev)
        b32 = ops.bn_backward(g32, x32, bits32, st, gamma, 3, want_g=True)
        ops.set_activation_dtype("bf16")
        x16, dy16, res16, g16 = r(x32), r(dy32), r(res32), r(g32)
        y16, p16 = ops.conv_fwd(x16, wp, Cout, K, K, s, p, True)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, r(y32))
        yr = y16.float().reshape(-1, Cout).double()
        sums = p16.double().sum(0)
        assert torch.allclose(sums[:, 0], yr.sum(0), rtol=0, atol=2e-6 * float(yr.abs().sum(0).max())) and torch.allclose(sums[:, 1], (yr * yr).sum(0), rtol=1e-5)
        assert torch.equal(ops.conv_bwd_data(dy16, wd, (N, H, W, C), Cout, K, K, s, p), r(dx32))
        acc16 = res16.clone()
        ops.conv_bwd_data(dy16, wd, (N, H, W, C), Cout, K, K, s, p, out=acc16, accumulate=True)
        assert torch.equal(acc16, r(acc32))
        dw16 = wgrad_both(x16, dy16)
        common = set(dw16) & set(dw32)
        assert common and all(torch.equal(dw16[k], dw32[k]) for k in common), (sorted(dw16), sorted(dw32))
        a16, bits16 = ops.bn_apply(x16, st, True, res16, want_bits=True)
        assert torch.equal(a16, r(a32)) and torch.equal(bits16, bits32)
        b16 = ops.bn_backward(g16, x16, bits16, st, gamma, 3, want_g=True)
        assert torch.equal(b16[0], r(b32[0])) and torch.equal(b16[1], r(b32[1])) and torch.equal(b16[2], b32[2]) and torch.equal(b16[3], b32[3])
    finally:
        ops.set_activation_dtype("fp32")
        ops.reset_compute_dtypes()


def test_bf16_activation_stem_and_pooling(dev, bf16_act):
    """stem (fp32 NHWC4 frames in, bf16 out), BatchNorm+ReLU+max-pool forward / backward and the pooled head on bf16 tensors vs the
    fp32-tensor instances of the same kernels fed the same bf16-representable values"""
    from litemkd_amd import ops
    torch.manual_seed(2)
    N, H = 4, 32
    frames = torch.rand(N, 3, H, H, device=dev)
    w = (torch.randn(64, 3, 7, 7, device=dev) * 0.1).requires_grad_()
    gamma, beta = (1 + 0.1 * torch.randn(64, device=dev)).requires_grad_(), (0.1 * torch.randn(64, device=dev)).requires_grad_()
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)

    def run():
        for t in (w, gamma, beta):
            t.grad = None
        y = ops.StemFn.apply(frames, w, gamma, beta, rm.clone(), rv.clone(), True)
        f = ops.PoolHeadFn.apply(y)
        f.backward(torch.linspace(-1, 1, f.numel(), device=dev).reshape(f.shape))
        return y.detach().float(), f.detach(), w.grad.clone(), gamma.grad.clone()
    y16, f16, dw16, dg16 = run()
    assert float(y16.abs().max()) > 0
    ops.set_activation_dtype("fp32")
    y32, f32, dw32, dg32 = run()
    # fp32 tensors keep the convolution output unrounded, so the two runs differ by bf16 rounding of the stored tensors (0.4 %)
    assert float((y16 - y32).abs().max()) <= 1e-2 * float(y32.abs().max())
    assert float((f16 - f32).abs().max()) <= 1e-2 * float(f32.abs().max())
    # gradients: the BatchNorm backward is a residual of large cancelling terms (g - mean g - xhat mean(g xhat)); with the gradient
    # tensors stored as bf16 (0.4 % per element) and 1 K samples per channel the weight gradient moves by several percent - the
    # nature of bf16 training tensors, not a kernel property.  Sanity bound here; the arithmetic itself is pinned by
    # test_episode_matches_oracle[...-bf16act] against the oracle that rounds the same tensors at the same places.
    assert float((dw16 - dw32).norm() / dw32.norm()) < 0.25 and float((dg16 - dg32).norm() / dg32.norm()) < 0.25


# ------------------------------------------------------------------------------------------
# same-size convolutions from an LDS-resident input patch (conv_patch.h)
# ------------------------------------------------------------------------------------------
PATCH_SHAPES = [  # N, C, H, W, Cout, K, pad
    (3, 64, 9, 11, 64, 3, 1),        # ragged: 297 pixels (not a multiple of any tile), odd width
    (2, 32, 5, 3, 96, 3, 1),         # fewer pixels than one tile, halo wider than a row, Cout not a multiple of the tile
    (1, 64, 1, 1, 32, 3, 1),         # a single pixel: eight of the nine taps are padding
    (5, 128, 14, 14, 256, 3, 1),     # several column tiles
    (2, 64, 56, 56, 64, 3, 1),       # the widest row the kernel takes (halo 57)
    (4, 256, 7, 7, 64, 1, 0),        # 1x1 / stride 1 (ResNet-50 bottleneck): halo 0
]


@pytest.mark.parametrize("tile", [0, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9", "bf16", "bf16act"])
def test_conv_patch_kernel_bit_identical_to_gather_kernel(dev, mode, tile):
    """conv_patch_x3_kernel multiplies the same bf16 planes in the same order as conv_gemm_x3_kernel (channel chunk outer, tap inner,
    one fp32 accumulator per output): forward, BatchNorm partial sums, data gradient, the accumulating data gradient and the fused
    BatchNorm+ReLU loader must agree BIT FOR BIT with the im2col-gather kernel, for every tile instance, on ragged shapes (pixel
    count below / not a multiple of the tile, one-pixel image, 1x1 convolution, widest supported row)."""
    import ctypes
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    act16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if mode.startswith("bf16") else mode)
    ops.set_activation_dtype("bf16" if act16 else "fp32")
    dt = torch.bfloat16 if act16 else torch.float32
    try:
        for (N, C, H, W, Cout, K, p) in PATCH_SHAPES:
            x = (rnd(N, H, W, C, seed=70 + H) * 1.5).to(dev).to(dt)
            w = (rnd(Cout, C, K, K, seed=71) * math.sqrt(2.0 / (Cout * K * K))).to(dev)
            dy = rnd(N, H, W, Cout, seed=72).to(dev).to(dt)
            r0 = rnd(N, H, W, C, seed=73).to(dev).to(dt)
            st = torch.zeros(5, C, device=dev)
            st[2], st[3] = (1 + 0.3 * rnd(C, seed=74)).to(dev), (0.2 * rnd(C, seed=75)).to(dev)
            wp, wd = ops._pack_weights(w, C, 0), ops._pack_weights(w, C, 1)
            out = {}
            # 0 = gather kernel, 1 = patch kernel on the 32x32x16 MFMA (bit-identical to the gather kernel), 2 = patch kernel on the
            # 16x16x32 MFMA (round 3: three-plane modes, tiles 11 / 12; 32 k per accumulation instead of 16: equal to fp32 rounding)
            for patch in (0, 1, 2):
                L.call("lmkd_conv_set_patch", min(patch, 1))
                L.call("lmkd_conv_set_patch16", 1 if patch == 2 else 0)
                L.call("lmkd_conv_set_tile", tile)
                info = (ctypes.c_int * 5)()
                L.call("lmkd_conv2d_plan", 0, N, H, W, C, C, Cout, K, K, 1, p, info)
                assert info[4] == min(patch, 1) and (tile == 0 or not patch or info[0] == (tile if Cout > 64 or tile in (7, 9, 11) else {8: 9, 10: 7, 12: 11}[tile])), list(info)
                y, part = ops.conv_fwd(x, wp, Cout, K, K, 1, p, True)
                dx = ops.conv_bwd_data(dy, wd, (N, H, W, C), Cout, K, K, 1, p)
                acc = r0.clone()
                ops.conv_bwd_data(dy, wd, (N, H, W, C), Cout, K, K, 1, p, out=acc, accumulate=True)
                o = [y, part.double().sum(0), dx, acc]
                if not act16:
                    o.append(ops.conv_fwd(x, wp, Cout, K, K, 1, p, True, pre_stats=st)[0])
                out[patch] = o
            for i, (g_, p_) in enumerate(zip(out[0], out[2])):      # the 16x16x32 kernel (where it exists; otherwise identical to out[1])
                tol = (1e-5 if i == 1 else 4e-6) * float(g_.float().abs().max()) + 1e-12
                assert float((g_.float() - p_.float()).abs().max()) <= tol, (mode, tile, (N, C, H, W, Cout, K), i, "16x16x32 kernel")
            for i, (g_, p_) in enumerate(zip(out[0], out[1])):
                if i == 1:      # per-row-tile partial sums (tile heights differ between instances): compare the column totals
                    assert torch.allclose(g_, p_, rtol=1e-6, atol=1e-6 * float(g_.abs().max()) + 1e-12), (mode, tile, (N, C, H, W, Cout, K), "BN sums")
                elif tile == 10 and mode.startswith("fp32x3"):
                    # 256-row tiles exist on the patch kernel only (the gather kernel runs 128-row tiles): in the three-plane modes the
                    # second HALF of a launch's row tiles accumulates -y (conv_x3.h, x3_neg_tile), so where the tile heights differ
                    # the two kernels add the same products with opposite accumulation sign on some rows - equal to fp32 rounding,
                    # not bit for bit
                    assert float((g_ - p_).abs().max()) <= 4e-6 * float(p_.abs().max()), (mode, tile, i)
                else:
                    assert torch.equal(g_, p_), (mode, tile, (N, C, H, W, Cout, K), i, float((g_.float() - p_.float()).abs().max()))
    finally:
        L.call("lmkd_conv_set_patch", 1)
        L.call("lmkd_conv_set_patch16", 1)
        L.call("lmkd_conv_set_tile", 0)
        ops.set_activation_dtype("fp32")
        ops.reset_compute_dtypes()


# ------------------------------------------------------------------------------------------
# 3x3 / stride-1 weight gradient from a rolling LDS window of x (wgrad_win.h)
# ------------------------------------------------------------------------------------------
WIN_SHAPES = [  # N, C, H, W, Cout
    (3, 64, 9, 11, 64),        # 297 pixels: the last step of a slab is ragged; 64 output channels (two-wave workgroups)
    (2, 32, 5, 3, 96),         # 30 pixels (one ragged step), halo wider than a row, Cout not a multiple of 128
    (1, 64, 1, 1, 32),         # a single pixel: eight of the nine taps are padding everywhere
    (5, 128, 14, 14, 256),     # several channel tiles and pixel splits
    (2, 64, 56, 56, 64),       # 56-wide rows: the 256-row ring
    (7, 64, 47, 5, 128),       # non-square, tall
]


@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9", "bf16", "bf16act"])
def test_wgrad_window_kernel_matches_gather_kernel_and_fp64(dev, mode):
    """conv_wgrad_win_kernel (all nine taps of a 3x3 from one rolling window of x rows in LDS, image borders as address selects on the
    per-row transposed reads) against conv_wgrad_x3_kernel (im2col gather) and an fp64 convolution weight gradient, on ragged shapes.
    Same arithmetic, different summation order: in the exact-split modes both must be within 3x of torch-CPU fp32's error against
    fp64; in the bf16 modes the two kernels must agree to fp32-accumulation level on identical bf16 operands."""
    import ctypes
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    act16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if mode.startswith("bf16") else mode)
    ops.set_activation_dtype("bf16" if act16 else "fp32")
    dt = torch.bfloat16 if act16 else torch.float32
    try:
        for (N, C, H, W, Cout) in WIN_SHAPES:
            x = (rnd(N, H, W, C, seed=80 + W) * 1.5).to(dev).to(dt)
            dy = rnd(N, H, W, Cout, seed=81).to(dev).to(dt)
            if mode.startswith("bf16"):      # both kernels round the same way: compare on bf16-representable operands
                x, dy = x.to(torch.bfloat16).to(dt), dy.to(torch.bfloat16).to(dt)
            out = {}
            for win in (0, 1):
                L.call("lmkd_conv_set_wgrad_window", win)
                info = (ctypes.c_int * 5)()
                L.call("lmkd_conv2d_plan", 2, N, H, W, C, C, Cout, 3, 3, 1, 1, info)
                assert info[4] == win, list(info)
                out[win] = ops.conv_bwd_weight(x, dy, (Cout, C, 3, 3), 1, 1)
            x64 = x.float().permute(0, 3, 1, 2).cpu().double()
            dy64 = dy.float().permute(0, 3, 1, 2).cpu().double()
            w64 = torch.zeros(Cout, C, 3, 3, dtype=torch.float64, requires_grad=True)
            torch.nn.functional.conv2d(x64, w64, None, 1, 1).backward(dy64)
            w32 = torch.zeros(Cout, C, 3, 3, requires_grad=True)
            torch.nn.functional.conv2d(x64.float(), w32, None, 1, 1).backward(dy64.float())
            ref = w64.grad
            e_cpu = float((w32.grad.double() - ref).norm() / ref.norm())
            for win in (0, 1):
                e = float((out[win].cpu().double() - ref).norm() / ref.norm())
                assert e <= 3 * e_cpu + 2e-6, (mode, win, (N, C, H, W, Cout), e, e_cpu)
            d = float((out[0].double() - out[1].double()).norm() / out[0].double().norm())
            assert d <= 3e-6, (mode, (N, C, H, W, Cout), d)
    finally:
        L.call("lmkd_conv_set_wgrad_window", 1)
        ops.set_activation_dtype("fp32")
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9", "bf16", "bf16act"])
def test_stem_patch_kernel_bit_identical_to_gather_kernel(dev, mode):
    """conv_stem_patch_kernel (7x7 / stride 2 from an LDS-resident patch of nine input rows, zero padding stored in the patch) sums the
    same products in the same order as the gather kernel (kernel row outer, (kw, c) inner): output and BatchNorm sums must agree bit
    for bit, also for an odd number of output rows (the last row pair is half empty), widths whose row pairs are not a whole number
    of 32-pixel blocks, and fewer than 64 output channels."""
    import ctypes
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    act16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if mode.startswith("bf16") else mode)
    ops.set_activation_dtype("bf16" if act16 else "fp32")
    try:
        for (N, H, W, Cout) in [(3, 64, 64, 64), (2, 58, 40, 64), (1, 21, 30, 32), (2, 224, 224, 64)]:
            x = torch.zeros(N, H, W, 4)
            x[..., :3] = rnd(N, H, W, 3, seed=90 + H)
            x = x.to(dev)
            w = (rnd(Cout, 3, 7, 7, seed=91) * math.sqrt(2.0 / (Cout * 49))).to(dev)
            wp = ops._pack_weights(w, 4, 0)
            out = {}
            for on in (0, 1):
                L.call("lmkd_conv_set_stem_patch", on)
                info = (ctypes.c_int * 5)()
                L.call("lmkd_conv2d_plan", 0, N, H, W, 4, 3, Cout, 7, 7, 2, 3, info)
                assert (info[4] == 2) == bool(on), list(info)
                y, part = ops.conv_fwd(x, wp, Cout, 7, 7, 2, 3, True)
                out[on] = (y, part.double().sum(0))
            assert torch.equal(out[0][0], out[1][0]), (mode, (N, H, W, Cout), float((out[0][0].float() - out[1][0].float()).abs().max()))
            assert torch.allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-6 * float(out[0][1].abs().max())), (mode, (N, H, W, Cout))
    finally:
        L.call("lmkd_conv_set_stem_patch", 1)
        ops.set_activation_dtype("fp32")
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("mode", ["fp32x3", "bf16", "bf16act"])
def test_stride2_data_gradient_on_patch_kernel_bit_identical(dev, mode):
    """a stride-2 data gradient is four same-size convolutions over the dy grid (one per input parity, 1-4 taps each) whose outputs
    are scattered to the pixels of their parity: the patch kernel runs them (plan info[4] == 1), bit-identical to the gather kernel,
    also through the accumulating epilogue, for the 1x1 / stride-2 downsample (three of its four classes have no tap) and for odd
    H, W (the last row / column of a class falls outside the image and is not stored)."""
    import ctypes
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    act16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if mode.startswith("bf16") else mode)
    ops.set_activation_dtype("bf16" if act16 else "fp32")
    dt = torch.bfloat16 if act16 else torch.float32
    try:
        for (N, Cin, H, W, Cout, K, p) in [(3, 64, 12, 20, 128, 3, 1), (2, 32, 56, 56, 64, 3, 1), (2, 64, 10, 6, 96, 1, 0), (5, 128, 14, 14, 256, 3, 1),
                                           (2, 64, 9, 11, 128, 3, 1)]:
            Ho, Wo = (H + 2 * p - K) // 2 + 1, (W + 2 * p - K) // 2 + 1
            w = (rnd(Cout, Cin, K, K, seed=95) * math.sqrt(2.0 / (Cout * K * K))).to(dev)
            dy = rnd(N, Ho, Wo, Cout, seed=96).to(dev).to(dt)
            r0 = rnd(N, H, W, Cin, seed=97).to(dev).to(dt)
            wd = ops._pack_weights(w, Cin, 1)
            out = {}
            for patch in (0, 1, 2):      # gather kernel, patch kernel on the 32x32x16 MFMA (bit-identical), on the 16x16x32 MFMA (fp32 rounding)
                L.call("lmkd_conv_set_patch", min(patch, 1))
                L.call("lmkd_conv_set_patch16", 1 if patch == 2 else 0)
                info = (ctypes.c_int * 5)()
                L.call("lmkd_conv2d_plan", 1, N, H, W, Cin, Cin, Cout, K, K, 2, p, info)
                assert info[4] == min(patch, 1) and info[2] == 4, list(info)
                dx = ops.conv_bwd_data(dy, wd, (N, H, W, Cin), Cout, K, K, 2, p)
                acc = r0.clone()
                ops.conv_bwd_data(dy, wd, (N, H, W, Cin), Cout, K, K, 2, p, out=acc, accumulate=True)
                out[patch] = (dx, acc)
            for i in range(2):
                assert torch.equal(out[0][i], out[1][i]), (mode, (N, Cin, H, W, Cout, K), i, float((out[0][i].float() - out[1][i].float()).abs().max()))
                d = float((out[0][i].float() - out[2][i].float()).abs().max())
                assert d <= 4e-6 * float(out[0][i].float().abs().max()), (mode, (N, Cin, H, W, Cout, K), i, d, "16x16x32 kernel")
    finally:
        L.call("lmkd_conv_set_patch", 1)
        L.call("lmkd_conv_set_patch16", 1)
        ops.set_activation_dtype("fp32")
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9"])
@pytest.mark.parametrize("cfg", [(40, 64, 56, 56, 128), (7, 128, 28, 28, 256), (3, 256, 14, 14, 512), (2, 64, 10, 6, 64), (5, 64, 112, 112, 64)])
def test_stride2_forward_on_patch_kernel(dev, cfg, mode):
    """conv_patch16_x3_kernel SRC2 (a stride-2 3x3 forward convolution as four same-size convolutions over the input's parity classes)
    against the im2col-gather kernel: same products, another summation order (taps class by class) - equal to fp32 rounding of the
    sums, BatchNorm partial sums likewise, and both within the anchored criterion of an fp64 convolution"""
    from litemkd_amd import ops
    from _anchor import anchored
    N, Cin, H, W, Cout = cfg
    x = torch.relu(rnd(N, H, W, Cin, seed=60)).to(dev)
    w = (rnd(Cout, Cin, 3, 3, seed=61) * math.sqrt(2.0 / (Cin * 9))).to(dev)
    ops.set_conv_compute_dtype(mode)
    try:
        wp = ops.pack_weights(w, Cin, 0)
        y1, p1 = ops.conv_fwd(x, wp, Cout, 3, 3, 2, 1, True)
        ops.lib().call("lmkd_conv_set_s2_patch", 0)
        y0, p0 = ops.conv_fwd(x, wp, Cout, 3, 3, 2, 1, True)
    finally:
        ops.lib().call("lmkd_conv_set_s2_patch", 1)
        ops.reset_compute_dtypes()
    assert y1.shape == y0.shape and p1.shape == p0.shape
    assert not torch.equal(y0, y1)          # the patch form did run
    scale = float(y0.abs().max())
    assert float((y1 - y0).abs().max()) <= 4e-6 * scale
    assert float((p1.sum(0) - p0.sum(0)).abs().max()) <= 1e-5 * float(p0.sum(0).abs().max())
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), w.cpu().double(), None, 2, 1).permute(0, 2, 3, 1)
    cpu = F.conv2d(x.cpu().permute(0, 3, 1, 2), w.cpu(), None, 2, 1).permute(0, 2, 3, 1)
    anchored("stride-2 forward, patch form", y1, cpu, ref, 3.0, 2e-6)


@pytest.mark.parametrize("mode", ["fp32x3", "bf16act"])
@pytest.mark.parametrize("N,H,W", [(6, 64, 64), (3, 62, 58), (2, 34, 70), (40, 224, 224)])
def test_stem_pooled_backward_matches_materialised_path(dev, N, H, W, mode):
    """Round 3: the stem's BatchNorm backward takes its sums from the POOLED tensors (dy, the convolution output at each window's
    arg-max: lmkd_bn_backward_stats) and forms dc in one pass that fuses the max-pool backward with the BatchNorm backward apply
    (lmkd_stem_unpool_bn_bwd, a thread per 2x2 block of pre-pooling pixels).  Against the round-2 path (maxpool_bwd materialises the
    pre-pooling gradient, bn_backward reduces and applies): the same selection of arg-max positions, the same arithmetic per
    element, sums in another order - weight / gamma / beta gradients agree to fp32 rounding.  Odd convolution-output sizes (31x29,
    17x35) exercise the block and window edges, 40 frames of 224^2 the full-size launch geometry."""
    from litemkd_amd import ops
    b16 = mode == "bf16act"
    ops.set_conv_compute_dtype("bf16" if b16 else mode)
    ops.set_activation_dtype("bf16" if b16 else "fp32")
    g = torch.Generator(device=dev).manual_seed(N * 1000 + H)
    x = torch.rand(N, 3, H, W, device=dev, generator=g)
    w = (torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.1).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(64, device=dev, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(64, device=dev, generator=g)).requires_grad_()
    # a few channels mostly negative after the BatchNorm: windows whose maximum is 0 (masked gradient)
    with torch.no_grad():
        beta[:8] -= 2.0
    res = {}
    for pooled in (True, False):
        ops.STEM_POOLED_BWD = pooled
        try:
            for t in (w, gamma, beta):
                t.grad = None
            rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
            y = ops.StemFn.apply(x, w, gamma, beta, rm, rv, True)
            gy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).to(y.dtype)
            y.backward(gy)
            res[pooled] = (y.detach().float().clone(), w.grad.clone(), gamma.grad.clone(), beta.grad.clone())
        finally:
            ops.STEM_POOLED_BWD = True
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])                                  # the forward does not change
    for name, u, v in (("dW", a[1], b[1]), ("dgamma", a[2], b[2]), ("dbeta", a[3], b[3])):
        tol = 2e-2 if b16 else 2e-5      # bf16 tensors: the old path rounds the pre-pooling gradient to bf16, the new one does not store it
        err = float((u - v).norm() / (v.norm() + 1e-30))
        assert err < tol, (name, err)
    assert float(a[3][:8].abs().max()) >= 0.0


def test_two_head_linear_matches_four_linear_calls(dev):
    """ops.TwoHeadLinearFn (fc1 / fc2 over both trunk calls' features as one autograd node, resnet18_2fc.py:56-64) against four
    LinearFn calls: outputs equal to fp32 rounding (the same GEMM rows; bit-identical on the native fp32 MFMA, while the 3 x bf16 GEMM
    alternates the accumulation sign per 32-row block of the launch - gemm_x3.h - and the query rows sit in other blocks of the stacked
    call), gradients equal to fp32 rounding of the re-ordered sums"""
    from litemkd_amd import ops
    from litemkd_amd.model.backbone.resnet import Linear
    torch.manual_seed(5)
    fc1, fc2 = Linear(512, 2048).to(dev), Linear(512, 2048).to(dev)
    cf0, tf0 = rnd(200, 512, seed=80).to(dev), rnd(120, 512, seed=81).to(dev)
    ws = [rnd(200, 2048, seed=82).to(dev), rnd(120, 2048, seed=83).to(dev), rnd(200, 2048, seed=84).to(dev), rnd(120, 2048, seed=85).to(dev)]
    res = {}
    for fused in (False, True):
        ops.FUSE_TWO_HEAD_LINEAR = fused
        try:
            for m in (fc1, fc2):
                m.zero_grad(set_to_none=True)
            cf, tf = cf0.clone().requires_grad_(), tf0.clone().requires_grad_()
            outs = ops.two_head_linear(cf, tf, fc1, fc2)
            sum((o * w).sum() for o, w in zip(outs, ws)).backward()
            res[fused] = [o.detach() for o in outs] + [cf.grad, tf.grad, fc1.weight.grad, fc1.bias.grad, fc2.weight.grad, fc2.bias.grad]
        finally:
            ops.FUSE_TWO_HEAD_LINEAR = True
    for i, (a, b) in enumerate(zip(res[False], res[True])):
        if i < 4:
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), i
        else:
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), (i, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("mode", ["fp32x3", "fp32x3_9", "bf16"])
def test_repack_multi_bit_identical_to_pack_and_split(dev, mode):
    """lmkd_conv2d_repack_multi (every cached pack of the convolution weights in one launch after an optimizer step) writes exactly
    the bytes lmkd_conv2d_pack_weights + lmkd_conv2d_split_weights write, for forward and data-gradient packs of every weight shape of
    the trunk (3x3, 1x1, the channel-padded stem)"""
    import ctypes
    from litemkd_amd import ops
    shapes = [(64, 3, 7, 7, 4, 0), (64, 64, 3, 3, 64, 0), (64, 64, 3, 3, 64, 1), (128, 64, 3, 3, 64, 0), (128, 64, 3, 3, 64, 1),
              (128, 64, 1, 1, 64, 0), (128, 64, 1, 1, 64, 1), (512, 512, 3, 3, 512, 0), (512, 512, 3, 3, 512, 1), (96, 32, 3, 3, 32, 0)]
    ops.set_conv_compute_dtype(mode)
    try:
        ws = [(rnd(co, ci, kh, kw, seed=90 + i) * 0.1).to(dev) for i, (co, ci, kh, kw, cs, md) in enumerate(shapes)]
        ref = [ops._pack_weights(w, s[4], s[5]) for w, s in zip(ws, shapes)]
        out = [torch.full_like(r, -1) for r in ref]
        m = len(shapes)
        a = (ctypes.c_void_p * m)(*[w.data_ptr() for w in ws])
        b = (ctypes.c_void_p * m)(*[o.data_ptr() for o in out])
        d = (ctypes.c_int * (6 * m))(*[v for s in shapes for v in (s[0], s[1], s[4], s[2], s[3], s[5])])
        ops.lib().call("lmkd_conv2d_repack_multi", a, b, d, m, ops._stream())
        for i, (r, o) in enumerate(zip(ref, out)):
            assert torch.equal(r, o), (mode, shapes[i], int((r != o).sum()))
    finally:
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("mode", ["fp32x3", "bf16"])
def test_trx_projections_on_conv_kernels_match_gemm(dev, mode):
    """ops.TRX_PROJ_ON_CONV (opt-in): the TRX projections and their input gradient as 1x1 convolutions in the three-plane arithmetic -
    also inside the one-plane bf16 mode, where the process-wide arithmetic is switched for those launches and restored (ops._x3_scope):
    logits and gradients equal to the GEMM path's to fp32-class rounding, and the mode is what it was afterwards"""
    from litemkd_amd import ops
    from litemkd_amd.model import classifiers as C
    args = _args(dev, shot=2)
    torch.manual_seed(11)
    clf = C.TRX_2fcsup(args).to(dev)
    sup = [rnd(10, 8, 2048, seed=100 + i).to(dev) for i in range(2)]
    qry = [rnd(5, 8, 2048, seed=110 + i).to(dev) for i in range(2)]
    lab = torch.arange(5).repeat_interleave(2).to(dev)
    ops.set_conv_compute_dtype(mode)
    if mode == "bf16":
        ops.set_activation_dtype("bf16")
    res = {}
    try:
        for on in (False, True):
            ops.TRX_PROJ_ON_CONV = on
            clf.zero_grad(set_to_none=True)
            t = [x.clone().requires_grad_() for x in sup + qry]
            r = clf({"context_features_1": t[0], "context_features_2": t[1]}, lab, {"target_features_1": t[2], "target_features_2": t[3]})["logits"]
            (r["kl"].square().sum() + r["ce"].square().sum()).backward()
            assert ops.get_conv_compute_dtype() == mode and ops.get_activation_dtype() == ("bf16" if mode == "bf16" else "fp32")
            res[on] = [r["kl"].detach(), r["ce"].detach()] + [x.grad for x in t] + [clf.transformers.k_linear.weight.grad.clone()]
    finally:
        ops.TRX_PROJ_ON_CONV = False
        ops.reset_compute_dtypes()
    for i, (a, b) in enumerate(zip(res[False], res[True])):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, (i, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize("layA,layB", [("K", "K"), ("K", "N"), ("M", "N"), ("M", "K")])
@pytest.mark.parametrize("M,N,K,batch", [(400, 1152, 2048, 2), (700, 140, 1152, 5), (1152, 2048, 400, 1), (100, 68, 36, 1), (64, 64, 4608, 1)])
def test_gemm_on_the_bf16_pipe(dev, layA, layB, M, N, K, batch):
    """lmkd_gemm_f32 in its three arithmetics (lmkd_gemm_set_mode; csrc/gemm_x3.h) against an fp64 product: the native fp32 MFMA kernel
    (mode 0), fp32 as 3 x bf16 with six products (mode 1: fp32-class error AND no coherent bias - the sign-alternating 32-row blocks
    cancel the MFMA's directional truncation), one bf16 plane per operand (mode 2: the fp64 product of the bf16-ROUNDED operands).
    All four operand layouts, ragged sizes (M, N not multiples of the tile, K not a multiple of 32), strided batches, alpha / beta / bias."""
    import litemkd_amd
    from litemkd_amd import ops
    lib = litemkd_amd.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(batch, M, K, generator=g) if layA == "K" else torch.randn(batch, K, M, generator=g)
    B = torch.randn(batch, N, K, generator=g) if layB == "K" else torch.randn(batch, K, N, generator=g)
    A = A + 0.5                                            # a mean, so that the sums are not centred on zero
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(batch, M, N, generator=g)
    A64 = (A if layA == "K" else A.transpose(1, 2)).double()
    B64 = (B if layB == "K" else B.transpose(1, 2)).double()
    r16 = lambda t: t.float().bfloat16().double()      # noqa: E731

    def ref(a, b):
        return 0.5 * (a @ b.transpose(1, 2)) + 0.25 * C0.double() + bias.double()
    Ad, Bd = A.to(dev), B.to(dev)
    lda, ldb = (K if layA == "K" else M), (K if layB == "K" else N)
    try:
        tiny = M * N * K < (1 << 18)      # such products stay on the fp32 MFMA kernel in every mode (gemm.hip gemm_planes)
        for mode, want in ((0, ref(A64, B64)), (1, ref(A64, B64)), (2, ref(A64, B64) if tiny else ref(r16(A64), r16(B64)))):
            lib.call("lmkd_gemm_set_mode", mode)
            C = C0.clone().to(dev)
            ops.gemm(layA, layB, M, N, K, Ad, lda, Bd, ldb, C, N, alpha=0.5, beta=0.25, bias=bias.to(dev), batch=batch,
                     sA=A[0].numel(), sB=B[0].numel(), sC=M * N)
            err = (C.cpu().double() - want)
            rel = float(err.norm() / want.norm())
            assert rel < 2e-6, (mode, rel)
            if mode == 1 and M * N >= 64 * 64 * 4:      # the coherent part of the error: far below the per-element error
                assert abs(float(err.mean())) < 0.1 * float(err.abs().mean()) + 1e-9, (float(err.mean()), float(err.abs().mean()))
    finally:
        lib.call("lmkd_gemm_set_mode", -1)
