"""GPU: the convolution kernels at the BENCHMARK's shapes (200 frames per trunk call, BASELINE configs[1]) — the kernel
instances, tile orders and epilogues bench.py times — against fp64 references, with an fp64-anchored criterion
(tests/_anchor.py): relative-L2 error vs fp64 <= 3 x the error of torch's own CPU fp32 convolution on the same data + a floor
of 1e-6 (forward, data gradient) / 2e-6 (weight gradient: sums over up to 2.5 M pixels).  torch-CPU's blocked summation reaches
1.4e-7 ... 2.5e-7 on the forward, so there the floor decides: the criterion reads "rel-L2 <= ~1.7e-6".  The measured
(HIP, CPU) pairs of a run are written to $LMKD_PARITY_LOG (committed as profiles/rNN_parity_errors.txt).

Forward and data gradient are per-frame independent, so they are checked on a frame subset (first and last frames: first
and last row tiles of the launch, i.e. both ends of every XCD band); the weight gradient sums over all 200 frames and is
checked in full.  lmkd_conv2d_plan reports which kernel instance / tile order each launch uses; the test asserts that the
shapes together exercise every instance the benchmark runs, and forces the remaining tile ids on one shape each."""
import ctypes
import zlib

import pytest
import torch
import torch.nn.functional as F

from _anchor import anchored

pytestmark = pytest.mark.gpu

FRAMES = 200
# name, H(=W), Cs (NHWC channels), Cin, Cout, K, stride, pad  — every distinct conv of the ResNet-18 trunk at 224x224
SHAPES = [
    ("stem7x7", 224, 4, 3, 64, 7, 2, 3),
    ("layer1.3x3", 56, 64, 64, 64, 3, 1, 1),
    ("layer2.0.conv1", 56, 64, 64, 128, 3, 2, 1),
    ("layer2.3x3", 28, 128, 128, 128, 3, 1, 1),
    ("layer2.ds", 56, 64, 64, 128, 1, 2, 0),
    ("layer3.0.conv1", 28, 128, 128, 256, 3, 2, 1),
    ("layer3.3x3", 14, 256, 256, 256, 3, 1, 1),
    ("layer3.ds", 28, 128, 128, 256, 1, 2, 0),
    ("layer4.0.conv1", 14, 256, 256, 512, 3, 2, 1),
    ("layer4.3x3", 7, 512, 512, 512, 3, 1, 1),
    ("layer4.ds", 14, 256, 256, 512, 1, 2, 0),
]
SEEN = {"gemm": set(), "wgrad": set(), "x3": set(), "x3w": set()}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import litemkd_amd
    litemkd_amd.lib().call("lmkd_device_check", 0)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    return torch.device("cuda", 0)


def plan(kind, N, H, Cs, Cin, Cout, K, s, p):
    import litemkd_amd
    info = (ctypes.c_int * 5)()
    litemkd_amd.lib().call("lmkd_conv2d_plan", kind, N, H, H, Cs, Cin, Cout, K, K, s, p, info)
    return tuple(info)


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _r16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def _ulp_close(name, hip16, ref64):
    """a bf16 tensor written by a kernel against the fp64 reference: equal to the ROUNDED reference except where the fp32
    accumulation lands within rounding of a bf16 rounding boundary (then one bf16 ulp away): every element within one ulp, at
    most 2 % of them different at all"""
    a = hip16.detach().cpu().double()
    r = _r16(ref64)
    # one bf16 ulp of |ref| (<= |ref| 2^-7) plus the fp32 accumulation noise of a K <= 4608 sum (elements that cancel to ~0)
    tol = ref64.abs() * 2.0 ** -7 + 4e-6 * float(ref64.abs().max())
    bad = (a - r).abs() > tol
    assert not bool(bad.any()), "%s: %d elements more than one bf16 ulp from the rounded fp64 reference" % (name, int(bad.sum()))
    frac = float((a != r).double().mean())
    assert frac < 0.02, "%s: %.2f %% of the elements differ from the rounded fp64 reference" % (name, 100 * frac)
    return frac


def _run_shape(dev, name, H, Cs, Cin, Cout, K, s, p, N=FRAMES, check_wgrad=True, act16=False, trained=False):
    """act16 (BASELINE configs[2]: ops.set_conv_compute_dtype('bf16') + set_activation_dtype('bf16') by the caller): x / dy / y / dx
    are bf16 tensors (the stem's NHWC4 input stays fp32), the kernels round the weights to one bf16 plane; the references are
    fp64 convolutions of the SAME bf16-rounded operands, so what is judged is the accumulation and the output rounding."""
    from litemkd_amd import ops
    from _anchor import record
    g = torch.Generator(device=dev).manual_seed(zlib.crc32(name.encode()) % 10007)
    x = torch.randn(N, H, H, Cs, device=dev, generator=g)
    if Cs != Cin:
        x[..., Cin:] = 0
    w = torch.randn(Cout, Cin, K, K, device=dev, generator=g) * (2.0 / (Cout * K * K)) ** 0.5
    Ho = (H + 2 * p - K) // s + 1
    dy = torch.randn(N, Ho, Ho, Cout, device=dev, generator=g)
    if trained:
        # statistics of a TRAINED network instead of the benchmark's fresh one (the reference starts from ImageNet weights,
        # resnet18_2fc.py:30): heavy-tailed pre-activations (log-normal magnitudes, sigma = 2), BatchNorm tables with |gamma| ~ U(0.2, 3),
        # either sign, beta ~ N(0, 1); weights whose output channels differ in scale by up to 2^6; gradients with log-normal magnitudes
        # and a 1e3 imbalance between the two halves of the frames (support / query) - inside ONE segment, i.e. under one scale
        sgn = lambda t: torch.where(t < 0, -torch.ones_like(t), torch.ones_like(t))      # noqa: E731
        gam = (0.2 + 2.8 * torch.rand(Cs, device=dev, generator=g)) * sgn(torch.randn(Cs, device=dev, generator=g))
        bet = torch.randn(Cs, device=dev, generator=g)
        t = torch.exp(2.0 * torch.randn(N, H, H, Cs, device=dev, generator=g)) * sgn(x)
        x = torch.relu(t * gam + bet)
        w = w * torch.exp2(6.0 * torch.rand(Cout, 1, 1, 1, device=dev, generator=g) - 3.0)
        dy = dy * torch.exp(torch.randn(N, Ho, Ho, Cout, device=dev, generator=g))
        dy[:N // 2] *= 1e3
    if act16:
        x = _r16(x)      # the values a bf16 tensor holds (the stem input, fp32 in HBM, is rounded by the kernel: same thing)
        dy = _r16(dy)
    xk = x.to(torch.bfloat16) if (act16 and Cs != 4) else x      # what the kernels are given
    dyk = dy.to(torch.bfloat16) if act16 else dy
    if ops.get_conv_compute_dtype() == "fp32h2":      # the operands' maxima (inside the trunk: recorded by the kernels that write them)
        ops.amax_compute(xk)
        ops.amax_compute(dyk)
    sub = list(range(4)) + list(range(N - 4, N))                 # first / last row tiles of the launch
    xs64 = nchw(x[sub][..., :Cin]).cpu().double()
    w64 = (_r16(w) if act16 else w).cpu().double()
    res = {}
    # ---- forward (+ BatchNorm partial sums in the epilogue)
    y, part = ops.conv_fwd(xk, ops.pack_weights(w, Cs, 0), Cout, K, K, s, p, True)
    ref = F.conv2d(xs64, w64, None, s, p)
    if act16:
        assert y.dtype == torch.bfloat16
        res["fwd != r(fp64)"] = (_ulp_close(name + " fwd", nchw(y[sub].float()), ref), 0.0)
    else:
        cpu = F.conv2d(xs64.float(), w64.float(), None, s, p)
        res["fwd"] = anchored(name + " fwd", nchw(y[sub]), cpu, ref, 3.0, 1e-6)
    sums = part.double().sum(0)
    yd = y.double().reshape(-1, Cout)
    # per-tile fp32 sums of the fp32 accumulators, combined in fp64: error relative to the sum of magnitudes
    assert torch.allclose(sums[:, 0], yd.sum(0), rtol=0, atol=2e-6 * float(yd.abs().sum(0).max())), name + " BN sum"
    assert torch.allclose(sums[:, 1], (yd * yd).sum(0), rtol=1e-5), name + " BN sum of squares"
    SEEN["gemm"].add(plan(0, N, H, Cs, Cin, Cout, K, s, p)[:2])
    SEEN["x3"].add(plan(0, N, H, Cs, Cin, Cout, K, s, p)[::4])      # (tile id, LDS-patch kernel?)
    # ---- data gradient (+ the accumulate epilogue at full size)
    if Cin != 3:
        wd = ops.pack_weights(w, Cin, 1)
        dx = ops.conv_bwd_data(dyk, wd, (N, H, H, Cin), Cout, K, K, s, p)
        dys64 = nchw(dy[sub]).cpu().double()
        xr = xs64.clone().requires_grad_()
        F.conv2d(xr, w64, None, s, p).backward(dys64)
        r = torch.randn(N, H, H, Cin, device=dev, generator=g)
        if act16:
            res["dgrad != r(fp64)"] = (_ulp_close(name + " dgrad", nchw(dx[sub].float()), xr.grad), 0.0)
            acc = r.to(torch.bfloat16)
            r64 = nchw(acc[sub].float()).cpu().double()
            ops.conv_bwd_data(dyk, wd, (N, H, H, Cin), Cout, K, K, s, p, out=acc, accumulate=True)
            _ulp_close(name + " dgrad accumulate", nchw(acc[sub].float()), r64 + xr.grad)
        else:
            xr32 = xs64.float().requires_grad_()
            F.conv2d(xr32, w64.float(), None, s, p).backward(dys64.float())
            res["dgrad"] = anchored(name + " dgrad", nchw(dx[sub]), xr32.grad, xr.grad, 3.0, 1e-6)
            acc = r.clone()
            ops.conv_bwd_data(dy, wd, (N, H, H, Cin), Cout, K, K, s, p, out=acc, accumulate=True)
            # out += dgrad: one fp32 add per element on top of the plain result
            assert float((acc - (r + dx)).abs().max()) <= 1e-6 * float(dx.abs().max()) + 1e-6 * float(r.abs().max()), name + " dgrad accumulate"
        SEEN["gemm"].add(plan(1, N, H, Cs, Cin, Cout, K, s, p)[:2])
    # ---- weight gradient over all frames (fp32 output in every mode; bf16 tensors: from the rounded x and dy)
    if check_wgrad:
        dw = ops.conv_bwd_weight(xk, dyk, (Cout, Cin, K, K), s, p)
        x64 = nchw(x[..., :Cin]).cpu().double()
        dy64 = nchw(dy).cpu().double()
        wr = w64.clone().requires_grad_()
        F.conv2d(x64, wr, None, s, p).backward(dy64)
        wr32 = w64.float().requires_grad_()
        F.conv2d(x64.float(), wr32, None, s, p).backward(dy64.float())
        res["wgrad"] = anchored(name + " wgrad", dw, wr32.grad, wr.grad)
        SEEN["wgrad"].add(plan(2, N, H, Cs, Cin, Cout, K, s, p)[:2])
        SEEN["x3w"].add(plan(2, N, H, Cs, Cin, Cout, K, s, p)[::4])      # (tile id, rolling-window kernel?)
    print(name, {k: "hip %.2e cpu %.2e" % v for k, v in res.items()})
    record("conv %s [%s%s%s]" % (name, ops.get_conv_compute_dtype(), ", bf16 tensors" if act16 else "", ", trained-like statistics" if trained else ""), res)
    return res


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_benchmark_shapes_vs_fp64(dev, shape):
    """native fp32 MFMA mode (lmkd_conv_set_compute_dtype(0))"""
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("fp32")
    _run_shape(dev, *shape)


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_benchmark_shapes_bf16_tensors(dev, shape):
    """BASELINE configs[2] at the benchmark's 200 frames: bf16 tensors in HBM, one-plane patch / window / gather / stem kernels
    (the kernels behind bench.py --dtype bf16) against fp64 convolutions of the same bf16-rounded operands: outputs equal to the
    rounded reference within one bf16 ulp, the fp32 weight gradient under the fp64-anchored criterion"""
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("bf16")
    ops.set_activation_dtype("bf16")
    _run_shape(dev, *shape, act16=True)
    pl = plan(0, FRAMES, shape[1], shape[2], shape[3], shape[4], shape[5], shape[6], shape[7])
    SEEN.setdefault("bf16_mode", set()).add((pl[0], pl[4]))


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_benchmark_shapes_x3_mode_vs_fp64(dev, shape):
    """the same launches in the fp32-as-3xbf16 arithmetic (conv_x3.h forward / data gradient, wgrad_x3.h weight gradient: exact
    3-way bf16 split of both operands, 6 bf16 MFMA products per fp32 product, fp32 accumulation) under the SAME fp64-anchored
    criterion as the native fp32 MFMA kernels: fp32-class error, not reduced precision"""
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("fp32x3")
    SEEN["x3"].clear()
    SEEN["x3w"].clear()
    try:
        _run_shape(dev, *shape)
        SEEN.setdefault("x3_mode", set()).update(SEEN["x3"])
        SEEN.setdefault("x3w_mode", set()).update(SEEN["x3w"])
    finally:
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_benchmark_shapes_h2_mode_vs_fp64(dev, shape):
    """bench.py's headline arithmetic ('fp32h2'): the 3x3 launches - forward, data gradient (+ its accumulate form), stride-1 weight
    gradient - on two fp16 planes with power-of-two scales, three v_mfma_f32_16x16x32_f16 products (csrc/conv_patch16.h), everything
    else in fp32x3; all under the SAME fp64-anchored criterion as the native fp32 MFMA kernels.  Asserts that the two-plane instances
    are what ran."""
    import litemkd_amd
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("fp32h2")
    n0 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    try:
        _run_shape(dev, *shape)
        ran = litemkd_amd.lib().value("lmkd_conv_h2_launches") - n0
        name, H, Cs, Cin, Cout, K, s, p = shape
        # 3x3: forward, data gradient, its accumulate form and the weight gradient (stride 1: the window kernel; stride 2: the gather
        # kernel); the 1x1 / stride-2 downsample convolution: its data gradient (+ accumulate form) runs on the patch kernel as four
        # parity classes with one tap, its weight gradient on the gather kernel, its forward stays in fp32x3; the stem: forward and weight
        # gradient on kernels of their own (conv_stem.h, wgrad_stem.h)
        want = (3 if (K == 1 and Cs % 32 == 0) else (2 if K == 7 else 0)) if K != 3 else 4
        assert ran == want, (name, ran, want)
    finally:
        ops.reset_compute_dtypes()


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[5] == 3], ids=[s[0] for s in SHAPES if s[5] == 3])
def test_conv_h2_trained_like_statistics_vs_fp64(dev, shape):
    """the headline arithmetic OUTSIDE the synthetic envelope: every 3x3 convolution of the trunk at 200 frames
    with trained-like tensors - log-normal activations (sigma = 2) through BatchNorm tables with |gamma| in [0.2, 3] of either sign and
    beta ~ N(0, 1), weights with a 2^6 spread of per-channel scales, log-normal gradients with a 1e3 support / query imbalance under ONE
    scale - under the same fp64-anchored criterion as the benchmark's tensors (3 x torch-CPU-fp32's own error + 1e-6 / 2e-6), with the
    two-plane launches asserted.  The pairs go to $LMKD_PARITY_LOG."""
    import litemkd_amd
    from litemkd_amd import ops
    ops.set_conv_compute_dtype("fp32h2")
    n0 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    try:
        _run_shape(dev, *shape, trained=True)
        assert litemkd_amd.lib().value("lmkd_conv_h2_launches") - n0 == 4, shape[0]
    finally:
        ops.reset_compute_dtypes()


def test_x3_benchmark_instances_were_exercised(dev):
    """in the headline arithmetic the shapes above must have run the LDS-patch kernel in both tiles the benchmark selects (11 =
    128x64 / 4 waves, 12 = 128x128 / 4 waves) and the im2col-gather kernel (stem, stride 2, 1x1: tiles 8 / 9)"""
    if not SEEN.get("x3_mode"):
        pytest.skip("run together with test_conv_benchmark_shapes_x3_mode_vs_fp64")
    assert {(11, 1), (12, 1)} <= SEEN["x3_mode"] and any(pt == 0 for _, pt in SEEN["x3_mode"]), SEEN["x3_mode"]
    # weight gradient: the rolling-window kernel with 128 (id 5) and 64 (id 6) output channels per workgroup, and the gather kernel
    assert {(5, 1), (6, 1)} <= SEEN["x3w_mode"] and any(w == 0 for _, w in SEEN["x3w_mode"]), SEEN["x3w_mode"]


@pytest.mark.parametrize("tile", [7, 8, 9, 10, 11, 12])
def test_patch_tile_instances_at_full_size(dev, tile):
    """every tile instance of conv_patch_x3_kernel (headline arithmetic) on a 200-frame layer (layer 1 for the 64-column tiles,
    layer 2 for the 128-column ones) under the fp64-anchored criterion, forward and data gradient"""
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    ops.set_conv_compute_dtype("fp32x3")
    L.call("lmkd_conv_set_tile", tile)
    try:
        H, C = (56, 64) if tile in (7, 9, 11) else (28, 128)
        pl = plan(0, FRAMES, H, C, C, C, 3, 1, 1)
        assert pl[0] == tile and pl[4] == 1, pl
        _run_shape(dev, "patch.tile%d" % tile, H, C, C, C, 3, 1, 1, check_wgrad=False)
    finally:
        L.call("lmkd_conv_set_tile", 0)
        ops.reset_compute_dtypes()


def test_benchmark_instances_were_exercised(dev):
    """the shapes above must have run both conv_gemm tiles the benchmark selects (64x64/4 waves = 3, 128x128/8 waves = 5) in
    both XCD tile orders, and the weight-gradient tiles 128x128, 128x64, 64x64 with and without XCD-grouped splits"""
    if len(SEEN["gemm"]) == 0:
        pytest.skip("run together with test_conv_benchmark_shapes_vs_fp64")
    assert {(3, 0), (3, 1), (5, 0)} <= SEEN["gemm"], SEEN
    assert {t for t, _ in SEEN["wgrad"]} >= {1, 2, 3} and {x for _, x in SEEN["wgrad"]} == {0, 1}, SEEN


@pytest.mark.parametrize("tile", [1, 2, 4, 6])
def test_other_tile_instances_at_full_size(dev, tile):
    """the conv_gemm instances the automatic choice does not pick at the benchmark shapes, forced on a 200-frame layer-2 conv"""
    import litemkd_amd
    from litemkd_amd import ops
    L = litemkd_amd.lib()
    ops.set_conv_compute_dtype("fp32")
    L.call("lmkd_conv_set_tile", tile)
    try:
        assert plan(0, FRAMES, 28, 128, 128, 128, 3, 1, 1)[0] == tile
        _run_shape(dev, "layer2.3x3.tile%d" % tile, 28, 128, 128, 128, 3, 1, 1, check_wgrad=False)
    finally:
        L.call("lmkd_conv_set_tile", 0)
