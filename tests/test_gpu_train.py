"""GPU: optimizer kernels against torch.optim, the episode loop against the oracle loop (A15, trainwandb.py:101-105,111-145),
and the TRX dropout path (A4, TRX_2fcsup.py:28,44-48)."""
import math

import pytest
import torch

from _anchor import anchored_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import litemkd_amd
    litemkd_amd.lib().call("lmkd_device_check", 0)
    return torch.device("cuda", 0)


class _Toy(torch.nn.Module):
    """parameter sizes that are not multiples of 4: FlatParams pads every parameter to a 16-byte boundary"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.a = torch.nn.Parameter(torch.randn(7, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(13, generator=g))
        self.c = torch.nn.Parameter(torch.randn(64, 3, 3, 3, generator=g))
        self.d = torch.nn.Parameter(torch.randn(1, generator=g))


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_fused_optimizer_matches_torch_optim(dev, opt):
    """lmkd_sgd_step / lmkd_adam_step on the flat buffers vs torch.optim.SGD / torch.optim.Adam (the optimizers the reference
    builds, trainwandb.py:101-104) on identical gradients, 7 steps with MultiStepLR milestones in between; the padding
    elements of the flat buffers stay zero."""
    from litemkd_amd import trainloop as TL
    lr = 0.05 if opt == "sgd" else 0.01
    m = _Toy().to(dev)
    ref = _Toy()
    fo = TL.FusedOptimizer(m, opt, lr)
    sch = TL.MultiStepLR(fo, [2, 5])
    to = (torch.optim.SGD if opt == "sgd" else torch.optim.Adam)(ref.parameters(), lr=lr)
    tsch = torch.optim.lr_scheduler.MultiStepLR(to, milestones=[2, 5], gamma=0.1)
    g = torch.Generator().manual_seed(9)
    pad_mask = torch.ones(fo.bucket.numel, dtype=torch.bool, device=dev)
    for p, o in zip(fo.bucket.params, fo.bucket.offsets):
        pad_mask[o:o + p.numel()] = False
    assert int(pad_mask.sum()) > 0
    for step in range(7):
        for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            gr = torch.randn(q.shape, generator=g) * (10.0 ** (step % 3 - 1))
            q.grad = gr.clone()
            p.grad.copy_(gr)
        fo.step()
        fo.zero_grad()
        to.step()
        to.zero_grad()
        sch.step()
        tsch.step()
        assert abs(fo.lr - to.param_groups[0]["lr"]) < 1e-12 * max(1.0, lr), (step, fo.lr, to.param_groups[0]["lr"])
        for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            err = float((p.detach().cpu() - q.detach()).abs().max())
            assert err <= 1e-6 * float(q.abs().max()) + 2e-6 * lr, (opt, step, n, err)
        assert float(fo.bucket.flat[pad_mask].abs().max()) == 0.0 and float(fo.bucket.grad.abs().max()) == 0.0


@pytest.mark.parametrize("opt,lr", [("sgd", 2e-3), ("adam", 1e-4)])
def test_episode_loop_matches_oracle_loop(dev, opt, lr):
    """trainloop.train on the real modules vs the oracle loop with torch.optim (trainwandb.py:101-105,111-145): 4 episodes,
    tasks_per_batch 3 -> optimizer steps after episodes 2 (two accumulated episodes) and 3 (iteration == total-1).  Losses of
    all episodes (the last two see updated weights), and the weight UPDATE of every tensor, fp64-anchored: error vs the oracle
    loop run in fp64 at most 3x the error of the oracle loop run in fp32."""
    from litemkd_amd import trainloop as TL
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd.utils import aggregate_accuracy
    from oracle import ref_cpu as O
    cfg = default_args(shot=1, query_per_class=1, img_size=64, trans_dropout=0.0, device=dev, training_iterations=4, tasks_per_batch=3,
                       learning_rate=lr, opt=opt, print_freq=100, save_freq=10 ** 9, sch=[2, 40000])
    torch.manual_seed(21)
    student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
    sp0 = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    tp = {k[len("classifier.transformers."):]: v.detach().cpu().clone() for k, v in teacher.state_dict().items()
          if k.startswith("classifier.transformers.")}
    eps = [O.make_episode(3100 + i, 5, 1, 1, img=64) for i in range(4)]
    fo = TL.FusedOptimizer(student, opt, lr)
    losses, accs = TL.train(student, teacher, [{k: v.unsqueeze(0) for k, v in e.items()} for e in eps],
                            Distiller(cfg.distill_name, cfg.cfg, dev), fo, TL.MultiStepLR(fo, cfg.sch), aggregate_accuracy, cfg)
    assert fo.steps == 2

    def oracle(dt):
        p = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sp0.items()}
        for k, v in p.items():
            if v.is_floating_point() and "running" not in k and not k.endswith("pe.pe"):
                v.requires_grad_()
        e = [{k: (v.to(dt) if v.is_floating_point() else v) for k, v in ep.items()} for ep in eps]
        lo, ac = O.train_loop(e, p, {k: v.to(dt) for k, v in tp.items()}, 5, 1, opt, lr, 3, (2, 40000), 4)
        return p, lo, ac
    p32, lo32, ac32 = oracle(torch.float32)
    p64, lo64, _ = oracle(torch.float64)
    # losses: the first two episodes see the initial weights (fp32 tolerance); the last two see weights after one / two optimizer
    # steps, where a loss is as sensitive to gradient rounding as the step is large: fp64-anchored like the weights
    for i in range(4):
        e_hip, e_cpu = abs(losses[i] - lo64[i]), abs(lo32[i] - lo64[i])
        assert e_hip <= 3 * e_cpu + 1e-3 * abs(lo64[i]), (i, losses, lo32, lo64)      # a scalar: error ratios of single numbers scatter
        if i < 2:
            assert abs(losses[i] - lo32[i]) < 1e-4 * max(1.0, abs(lo32[i])), (i, losses, lo32)
    names = [k for k, v in p64.items() if v.is_floating_point() and v.requires_grad]
    d64 = {k: (p64[k].detach() - sp0[k].double()) for k in names}
    # Adam turns an exactly-zero gradient (biases that cancel in q - s differences: fp32 rounding noise vs eps = 1e-8) into an
    # update of rounding-noise sign: those tensors are compared through their gradient-free invariants only
    dmax = max(float(v.abs().max()) for v in d64.values())
    names = [k for k in names if float(d64[k].abs().max()) > 1e-4 * dmax]
    assert len(names) > 55
    sd = student.state_dict()
    hip = {k: sd[k].detach().cpu().double() - sp0[k].double() for k in names}
    c32 = {k: p32[k].detach().double() - sp0[k].double() for k in names}
    if opt == "sgd":
        worst = anchored_dict(hip, c32, {k: d64[k] for k in names}, floor=5e-6)
    else:
        # Adam's first update is lr * g / (|g| + eps) ~ lr * sign(g): an element whose gradient is rounding noise around zero moves
        # by +-lr in every fp32 evaluation (one such element in a 64-element tensor is a 25 % relative-L2 "error" against fp64).
        # So per tensor: the FRACTION of elements whose update is off by more than 0.2 lr, again anchored on the CPU fp32 run.
        # (The kernel itself is compared exactly with torch.optim.Adam on identical gradients above.)
        worst = (0.0, "")
        for k in names:
            bad_hip = float(((hip[k] - d64[k]).abs() > 0.2 * lr).double().mean())
            bad_cpu = float(((c32[k] - d64[k]).abs() > 0.2 * lr).double().mean())
            assert bad_hip <= 3 * bad_cpu + 0.05, (k, bad_hip, bad_cpu)
            worst = max(worst, (bad_hip, k))
    # BatchNorm running statistics after 8 trunk calls (support, query per episode): order and momentum as the reference
    for k in ("backbone.resnet.1.running_mean", "backbone.resnet.7.1.bn2.running_var", "backbone.resnet.5.0.downsample.1.running_var"):
        a, b, r = sd[k].cpu().double(), p32[k].double(), p64[k]
        e_hip, e_cpu = float((a - r).abs().max()), float((b - r).abs().max())
        assert e_hip <= 3 * e_cpu + 1e-4 * float(r.abs().max()), (k, e_hip, e_cpu)
    assert int(sd["backbone.resnet.1.num_batches_tracked"]) == 8
    print(opt, "losses", losses, lo32, "worst update error ratio", worst)


def test_dropout_mask_properties(dev):
    """hash-RNG dropout mask (lmkd_dropout_mask): values in {0, 1/(1-p)}, keep rate within 3 sigma of 1-p, deterministic per
    seed, independent across seeds"""
    from litemkd_amd import ops
    n, p = 400 * 2048, 0.1
    m1 = ops.dropout_mask((400, 2048), p, 12345, dev)
    m2 = ops.dropout_mask((400, 2048), p, 12345, dev)
    m3 = ops.dropout_mask((400, 2048), p, 12346, dev)
    assert torch.equal(m1, m2)
    vals = torch.unique(m1).cpu()
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / 0.9) < 1e-6, vals
    sigma = math.sqrt(p * (1 - p) / n)
    for m in (m1, m3):
        keep = float((m > 0).double().mean())
        assert abs(keep - (1 - p)) < 3 * sigma, keep
        rows = (m > 0).double().mean(1)            # no structure along rows / columns
        cols = (m > 0).double().mean(0)
        assert float((rows - 0.9).abs().max()) < 6 * math.sqrt(0.09 / 2048) and float((cols - 0.9).abs().max()) < 6 * math.sqrt(0.09 / 400)
    both = float(((m1 > 0) & (m3 > 0)).double().mean())                 # independent masks: P(both kept) = 0.81
    assert abs(both - 0.81) < 4 * math.sqrt(0.81 * 0.19 / n), both


def test_trx_dropout_forward_and_backward_use_the_same_mask(dev):
    """TemporalCrossTransformer in train mode, trans_dropout 0.1 (the benchmark's and the reference's setting).  With the mask
    m the module drew (reproduced from the torch RNG state), dropout(x + pe) = (x + pe) * m, so the train-mode logits on x must
    equal the eval-mode logits on x' = (x + pe) * m - pe, and d loss / d x = m * d loss / d x' (kept values scaled by 1/0.9,
    dropped positions get exactly zero gradient): forward and backward apply the identical mask."""
    from litemkd_amd import ops
    from litemkd_amd.model.classifiers.TRX_2fcsup import TemporalCrossTransformer
    from litemkd_amd.options import default_args
    args = default_args(shot=2, query_per_class=1, trans_dropout=0.1, device=dev)
    torch.manual_seed(5)
    t = TemporalCrossTransformer(args).to(dev).train()
    g = torch.Generator().manual_seed(6)
    sup = (torch.randn(10, 8, 2048, generator=g) * 0.5).to(dev)
    qry = (torch.randn(5, 8, 2048, generator=g) * 0.5).to(dev)
    lab = torch.arange(5).repeat_interleave(2)[torch.randperm(10, generator=g)].float().to(dev)
    w = torch.linspace(-1, 1, 25, device=dev).reshape(5, 5)
    state = torch.get_rng_state()
    s1, q1 = sup.clone().requires_grad_(), qry.clone().requires_grad_()
    l1 = t(s1, lab, q1)["logits"]
    (l1 * w).sum().backward()
    gk = t.k_linear.weight.grad.clone()
    t.zero_grad()
    torch.set_rng_state(state)
    m = ops.dropout_mask((15 * 8, 2048), 0.1, TemporalCrossTransformer.draw_dropout_seed(), dev)
    keep = float((m > 0).double().mean())
    assert abs(keep - 0.9) < 3 * math.sqrt(0.09 / m.numel())
    ms, mq = m[:80].reshape(10, 8, 2048), m[80:].reshape(5, 8, 2048)
    pe = t.pe.pe[0, :8]
    t.eval()
    s2 = ((sup + pe) * ms - pe).requires_grad_()
    q2 = ((qry + pe) * mq - pe).requires_grad_()
    l2 = t(s2, lab, q2)["logits"]
    (l2 * w).sum().backward()
    assert float((l1 - l2).abs().max()) <= 2e-5 * float(l2.abs().max()), float((l1 - l2).abs().max())
    for a, b, mm in ((s1.grad, s2.grad, ms), (q1.grad, q2.grad, mq)):
        assert float(a[mm == 0].abs().max()) == 0.0                       # dropped positions: exactly zero gradient
        assert float((a - b * mm).abs().max()) <= 2e-5 * float(b.abs().max())
    assert float((gk - t.k_linear.weight.grad).abs().max()) <= 2e-5 * float(gk.abs().max())
    # a second train-mode call draws a new mask
    t.train()
    l3 = t(sup, lab, qry)["logits"]
    assert float((l3 - l1).abs().max()) > 0


@pytest.mark.parametrize("conv", ["fp32x3", "fp32h2"])
def test_data_parallel_world2_equals_world1(dev, tmp_path, conv):
    """Episode parallelism (parallel.py): 16 global episodes dealt to 2 ranks (two processes on this one GPU, gloo
    — the bucket / all-reduce / optimizer code is the one RCCL runs), one all-reduce of the flat gradient bucket, one SGD
    step: the updated weights equal the 1-process run over the same 16 episodes to fp32 rounding (different summation
    order of the 16 episode gradients only).  The checkpoint holds the rank-pooled BatchNorm running statistics."""
    import glob
    import os
    import socket
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")

    def launch(world):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       LMKD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", LMKD_CONV=conv)
            procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), "16"], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
    launch(1)
    launch(2)
    a, b = torch.load(str(tmp_path / "flat_w1.pt")), torch.load(str(tmp_path / "flat_w2.pt"))
    assert torch.equal(a["w0"], b["w0"])
    d1, d2 = (a["w1"] - a["w0"]).double(), (b["w1"] - b["w0"]).double()
    assert float(d1.norm()) > 0
    assert float((d1 - d2).norm() / d1.norm()) < 1e-5, float((d1 - d2).norm() / d1.norm())
    assert float((d1 - d2).abs().max()) <= 1e-5 * float(d1.abs().max())
    # checkpoint of the 2-rank run: running statistics = mean over the ranks' own estimates
    r0, r1 = torch.load(str(tmp_path / "bn_w2_r0.pt")), torch.load(str(tmp_path / "bn_w2_r1.pt"))
    ck = torch.load(glob.glob(str(tmp_path / "*w2_1.pt"))[0])["model_state_dict"]
    for k in r0:
        if k.endswith("running_mean"):
            assert torch.allclose(ck[k], (r0[k] + r1[k]) / 2, rtol=1e-6, atol=1e-7), k
            kv = k[:-len("running_mean")] + "running_var"      # pooled variance: mean of the variances + spread of the means
            pooled = (r0[kv] + r1[kv]) / 2 + (r0[k] ** 2 + r1[k] ** 2) / 2 - ((r0[k] + r1[k]) / 2) ** 2
            assert torch.allclose(ck[kv], pooled, rtol=1e-5, atol=1e-6), kv
    assert not torch.equal(r0["backbone.resnet.1.running_mean"], r1["backbone.resnet.1.running_mean"])


@pytest.mark.parametrize("fresh_tensors,conv", [(False, "fp32x3"), (True, "fp32x3"), (False, "fp32h2"), (True, "fp32h2")])
def test_graphed_episode_bit_identical_to_eager(dev, fresh_tensors, conv):
    """trainloop.GraphedEpisode: the episode captured into a hipGraph (forward + loss + backward on three streams, dropout seeds and
    class plan in device memory, packed weights refreshed in place) replays the SAME kernels on the same data: losses, accuracies,
    accumulated gradients, BatchNorm running statistics and the weights after an optimizer step are bit-identical to the eager loop
    over the same episodes with the same RNG state (dropout 0.1 active).
    fresh_tensors False: a resident pool of episodes - one graph per episode, captured on the episode's own tensors (bench.py);
    True: every episode arrives in new tensors (a data loader) - ONE graph on static copies, inputs and class plan copied in.
    conv fp32h2 (bench.py's headline arithmetic): the words that carry the tensors' maxima are zeroed INSIDE the graph (ops._amax_slot)."""
    from litemkd_amd import ops, trainloop as TL
    ops.set_conv_compute_dtype(conv)
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd.utils import aggregate_accuracy
    cfg = default_args(shot=1, query_per_class=1, img_size=64, trans_dropout=0.1, device=dev, learning_rate=1e-2)
    src = TL.SyntheticEpisodes(cfg, base_seed=77, device=dev)
    pool = [src.episode(e) for e in range(2)]
    order = [0, 1, 0, 1, 0, 1, 1, 0]

    def run(graph):
        torch.manual_seed(91)
        student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
        opt = TL.FusedOptimizer(student, "sgd", cfg.learning_rate)
        distiller = Distiller(cfg.distill_name, cfg.cfg, dev)
        runner = TL.GraphedEpisode(student, teacher, distiller, aggregate_accuracy, cfg, max_graphs=2) if graph else None
        prev = (ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD)
        ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
        out = []
        try:
            torch.manual_seed(5)                      # the dropout seeds are drawn from the CPU generator, one per head and episode
            for i, e in enumerate(order):
                if graph:
                    ep = {k: v.clone() for k, v in pool[e].items()} if fresh_tensors else pool[e]
                    loss, acc, _ = runner(ep)
                else:
                    loss, acc, _ = TL.train_task(pool[e], student, teacher, distiller, aggregate_accuracy, cfg)
                out.append((float(loss), float(acc)))
                if i == 4:                              # an optimizer step in the middle: the packed weights must follow
                    opt.step()
                    opt.zero_grad()
            ops.wait_weight_grads()
            opt.bucket.fold_shadow()
            torch.cuda.synchronize()
        finally:
            ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = prev
        stats = {k: v.clone() for k, v in student.state_dict().items() if "running" in k or "num_batches" in k}
        return out, opt.bucket.grad.clone(), opt.bucket.flat.clone(), stats, runner
    try:
        o_e, g_e, w_e, s_e, _ = run(False)
        o_g, g_g, w_g, s_g, runner = run(True)
    finally:
        ops.reset_compute_dtypes()
    if fresh_tensors:      # two new keys eager (max_graphs), then the generic graph: its first episode eager, captured on the next
        assert runner.replays == len(order) - 3 and runner.eager == 3 and len(runner.graphs) == 1, (runner.replays, runner.eager)
    else:                  # first sight of each of the two episodes eager, then their own graphs
        assert runner.replays == len(order) - 2 and runner.eager == 2 and len(runner.graphs) == 2, (runner.replays, runner.eager)
    assert o_e == o_g, (o_e, o_g)
    assert torch.equal(w_e, w_g)
    assert torch.equal(g_e, g_g), float((g_e - g_g).abs().max())
    for k in s_e:
        assert torch.equal(s_e[k], s_g[k]), k
    assert len({l for l, _ in o_e}) > 4                 # the episodes really differ (dropout masks, optimizer step)


def test_graphed_interval_bit_identical_to_eager(dev):
    """GraphedEpisode.run_interval: the episodes between two optimizer steps captured as ONE hipGraph (the weight-gradient stream joins
    once, at the end of the interval): losses, accumulated gradients, weights after the steps and BatchNorm running statistics
    bit-identical to the eager loop; the interval is seen once eagerly, captured at its second occurrence, replayed afterwards"""
    from litemkd_amd import ops, trainloop as TL
    from litemkd_amd.model.model_select import Student, Teacher
    from litemkd_amd.distillers import Distiller
    from litemkd_amd.options import default_args
    from litemkd_amd.utils import aggregate_accuracy
    cfg = default_args(shot=1, query_per_class=1, img_size=64, trans_dropout=0.1, device=dev, learning_rate=1e-2)
    src = TL.SyntheticEpisodes(cfg, base_seed=78, device=dev)
    pool = [src.episode(e) for e in range(2)]
    interval = [0, 1, 0]
    n_int = 4

    def run(graph):
        torch.manual_seed(92)
        student, teacher = Student(cfg).to(dev), Teacher(cfg).to(dev)
        opt = TL.FusedOptimizer(student, "sgd", cfg.learning_rate)
        distiller = Distiller(cfg.distill_name, cfg.cfg, dev)
        runner = TL.GraphedEpisode(student, teacher, distiller, aggregate_accuracy, cfg) if graph else None
        prev = (ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD)
        ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
        out = []
        try:
            torch.manual_seed(6)
            for it in range(n_int):
                eps = [pool[e] for e in interval]
                res = runner.run_interval(eps) if graph else [TL.train_task(e, student, teacher, distiller, aggregate_accuracy, cfg) for e in eps]
                out += [(float(l), float(a)) for l, a, _ in res]
                if it < n_int - 1:
                    opt.step()
                    opt.zero_grad()
            ops.wait_weight_grads()
            opt.bucket.fold_shadow()
            torch.cuda.synchronize()
        finally:
            ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = prev
        stats = {k: v.clone() for k, v in student.state_dict().items() if "running" in k or "num_batches" in k}
        return out, opt.bucket.grad.clone(), opt.bucket.flat.clone(), stats, runner
    o_e, g_e, w_e, s_e, _ = run(False)
    o_g, g_g, w_g, s_g, runner = run(True)
    assert runner.eager == len(interval) and runner.replays == (n_int - 1) * len(interval) and len(runner.graphs) == 1, (runner.eager, runner.replays)
    assert o_e == o_g, (o_e, o_g)
    assert torch.equal(w_e, w_g)
    assert torch.equal(g_e, g_g), float((g_e - g_g).abs().max())
    for k in s_e:
        assert torch.equal(s_e[k], s_g[k]), k


def test_pipelined_episodes_bit_identical_to_sequential(dev):
    """trainloop.PipelinedEpisodes: the forward of episode i + 1 queued beside the backward of episode i on a second stream set.
    Same kernels on the same data, forwards and backwards each in program order: losses, accuracies, the accumulated flat gradient
    (incl. the side-stream shadow), BatchNorm running statistics and the weights after the optimizer steps are BIT-identical to the
    sequential loop (dropout 0.1 active, same RNG state, an optimizer step in the middle and one at the end)."""
    from litemkd_amd import ops, trainloop as TL
    from litemkd_amd.options import default_args
    cfg = default_args(shot=1, query_per_class=1, img_size=64, trans_dropout=0.1, device=dev, learning_rate=1e-2)
    src = TL.SyntheticEpisodes(cfg, base_seed=177, device=dev)
    eps = [src.episode(e) for e in range(7)]

    def run(pipelined):
        torch.manual_seed(91)
        student, teacher, _, distiller, accuracy_fn, _, opt, _ = TL.make(cfg)
        prev = (ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD)
        ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = True, False, True
        out = []
        try:
            torch.manual_seed(5)
            pipe = TL.PipelinedEpisodes(student, teacher, distiller, accuracy_fn, cfg) if pipelined else None
            for i, ep in enumerate(eps):
                if pipelined:
                    r = pipe.push(ep)
                    if r is not None:
                        out.append(r)
                else:
                    loss, acc, _ = TL.train_task(ep, student, teacher, distiller, accuracy_fn, cfg)
                    out.append((loss, acc))
                if i == 3:
                    if pipelined:
                        out.append(pipe.flush())
                    opt.step()
                    opt.zero_grad()
            if pipelined:
                out.append(pipe.flush())
            ops.wait_weight_grads()
            opt.bucket.fold_shadow()
            torch.cuda.synchronize()
        finally:
            ops.SIDE_WGRAD, ops.SYNC_WGRAD_AT_BACKWARD_END, ops.DIRECT_PARAM_GRAD = prev
        stats = {k: v.clone() for k, v in student.state_dict().items() if "running" in k or "num_batches" in k}
        return [(float(l), float(a)) for l, a in out], opt.bucket.grad.clone(), opt.bucket.flat.clone(), stats
    o_s, g_s, w_s, s_s = run(False)
    o_p, g_p, w_p, s_p = run(True)
    assert len(o_p) == len(eps) and o_s == o_p, (o_s, o_p)
    assert torch.equal(w_s, w_p)
    assert torch.equal(g_s, g_p), float((g_s - g_p).abs().max())
    for k in s_s:
        assert torch.equal(s_s[k], s_p[k]), k


class _HostEpisodes:
    """a host-side episode source for trainloop.StreamedEpisodes: n seeded episodes of decoded uint8 frames (one resolution), crop / flip
    parameters, teacher features and shuffled labels, handed out in order"""

    def __init__(self, cfg, n, seed=5):
        import random
        from litemkd_amd.video_transform import GpuFrameTransform
        g = torch.Generator().manual_seed(seed)
        rnd = random.Random(seed)
        tf = GpuFrameTransform(cfg.img_size, "cpu")
        ns, nq, L = cfg.way * cfg.shot, cfg.way * cfg.query_per_class, cfg.seq_len
        self.eps, self.i, self.length = [], 0, n
        state = random.getstate()
        random.seed(seed)
        for _ in range(n):
            self.eps.append({"frames": torch.randint(0, 256, ((ns + nq) * L, 100, 132, 3), dtype=torch.uint8, generator=g),
                             "params": [tf.draw(100, 132, True) for _ in range(ns + nq)],
                             "features": torch.randn(ns + nq, L, 2048, generator=g), "ns": ns,
                             "support_labels": torch.arange(cfg.way).repeat_interleave(cfg.shot)[torch.randperm(ns, generator=g)].float(),
                             "target_labels": torch.arange(cfg.way).repeat_interleave(cfg.query_per_class)[torch.randperm(nq, generator=g)].float()})
        random.setstate(state)
        del rnd

    def host_episode(self):
        e = self.eps[self.i % len(self.eps)]
        self.i += 1
        return e


def test_streamed_loader_equals_resident_inputs(dev):
    """trainloop.StreamedEpisodes (the loader inside the product loop, under the schedule that is timed): a
    prefetch thread hands over pinned host episodes, a copy stream uploads the uint8 frames + features and runs the frame transform into
    one of three static input sets, the pipelined loop (Schedule.bench) consumes them and releases each set to the copy stream through
    events.  Seven episodes with optimizer steps: losses, accuracies and the weights are BIT-identical to the same loop fed with the same
    episodes transformed up front on the compute stream (resident inputs)."""
    from litemkd_amd import trainloop as TL
    from litemkd_amd.options import default_args
    from litemkd_amd.schedule import Schedule
    from litemkd_amd.video_transform import GpuFrameTransform
    n = 7
    cfg = default_args(shot=1, query_per_class=1, img_size=84, trans_dropout=0.1, device=dev, learning_rate=1e-2, training_iterations=n,
                       tasks_per_batch=4, print_freq=10 ** 9, save_freq=10 ** 9)

    def resident():
        src, tf = _HostEpisodes(cfg, n), GpuFrameTransform(cfg.img_size, dev)
        L, out = cfg.seq_len, []
        for h in src.eps:
            x = tf.batch(h["frames"].to(dev), h["params"], L)
            ns, f = h["ns"], h["features"].to(dev)
            d = {"support_set": x[:ns * L], "target_set": x[ns * L:], "support_set_feature_teacher": f[:ns], "target_set_feature_teacher": f[ns:],
                 "support_labels": h["support_labels"], "target_labels": h["target_labels"]}
            out.append({k: v.unsqueeze(0) for k, v in d.items()})
        return out

    def run(streamed):
        sched = Schedule.bench(conv_dtype="fp32h2")
        torch.manual_seed(17)
        with sched.applied():
            student, teacher, _, distiller, acc_fn, _, opt, sch = TL.make(cfg)
            loader = TL.StreamedEpisodes(_HostEpisodes(cfg, n), cfg, dev) if streamed else resident()
            torch.manual_seed(3)
            losses, accs = TL.train(student, teacher, loader, distiller, opt, sch, acc_fn, cfg, schedule=sched)
            torch.cuda.synchronize()
            if streamed:
                assert loader.staged == n and all(s is not None for s in loader.sets)      # three sets, each staged more than once
                loader.close()
        return losses, accs, opt.bucket.flat.clone()
    l_r, a_r, w_r = run(False)
    l_s, a_s, w_s = run(True)
    assert len(l_s) == n and l_s == l_r and a_s == a_r, (l_s, l_r)
    assert torch.equal(w_s, w_r)


def test_stream_lifetime_audit_finds_nothing(dev):
    """lite-mkd_amd/_audit.py: every entry-point call of seven 64-px training episodes with optimizer steps under
    Schedule.bench() and Schedule.two_call() in fp32h2, with each tensor argument checked - a tensor used on a stream other than the one it
    was allocated on must carry a record_stream for that stream (or a declared reason).  In a process of its own: the audit wraps torch's
    allocation functions at import."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LMKD_STREAM_AUDIT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_audit.py"), "64"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "total findings 0" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    # negative control: with the record_stream of the maxima's words withheld (round 4's bug) the audit names the weight gradient
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_audit.py"), "64", "--selftest"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "total findings 0" not in r.stdout and "lmkd_conv2d_bwd_weight_seg" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


def test_rccl_world2_when_two_gpus_are_visible(dev, tmp_path):
    """RCCL readiness.  Needs >= 2 visible GPUs - skips on the 1-GPU test boxes, so the suite stays green - and
    then runs the REAL `nccl` (= RCCL) backend twice: (a) the data-parallel worker of test_data_parallel_world2_equals_world1 with one GPU
    per rank - world-2 weights after the all-reduce + SGD step equal the world-1 run, the early all-reduce of the bucket's tail fires;
    (b) `python bench.py --gpus 2` through its own launcher (bench.self_launch: torchrun as a child process, started before any GPU call)
    - the JSON line reports backend nccl, two distinct devices, a correct all-reduce checksum and the optimizer step's collective."""
    import json
    import os
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL needs one device per rank")
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "_dp_worker.py")

    def launch(world):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       LMKD_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), "16"], env=env))
        for p in procs:
            assert p.wait(timeout=900) == 0
    launch(1)
    launch(2)
    a, b = torch.load(str(tmp_path / "flat_w1.pt")), torch.load(str(tmp_path / "flat_w2.pt"))
    assert torch.equal(a["w0"], b["w0"])
    d1, d2 = (a["w1"] - a["w0"]).double(), (b["w1"] - b["w0"]).double()
    assert float(d1.norm()) > 0 and float((d1 - d2).norm() / d1.norm()) < 1e-5
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(here), "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "2",
                          "--no-cpu-baseline", "--no-other-modes"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    d = line["distributed"]
    assert line["n_gpus"] == 2 and d["backend"] == "nccl" and d["world"] == 2 and d["allreduce_checksum_ok"]
    assert len(set(d["devices"])) == 2 and d["optimizer_steps_in_timed_region"] >= 1
    assert d["allreduce_ms_per_optimizer_step"] > 0 and d["allreduce_early_elements"] > 0


def test_bench_world2_gloo_rehearsal_on_one_gpu(dev):
    """the whole N > 1 path of bench.py on ONE card: two ranks over gloo (LMKD_DIST_BACKEND=gloo lets ranks share a GPU; RCCL cannot), started the
    way the driver starts it (torch.distributed.run).  Round 5 found a hang-in-waiting here: the allocator-priming loop in front of the warm-up left on a
    PER-RANK condition, so two ranks could run a different number of optimizer intervals - each of which holds the step's all-reduces.  The exit is
    collective now: both ranks report the same priming count, the timed region has its optimizer step, the checksum of the all-reduced bucket agrees."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LMKD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--prime", "48",
                          "--no-cpu-baseline", "--no-other-modes"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    line = json.loads(lines[0])
    d = line["distributed"]
    assert line["n_gpus"] == 2 and d["backend"] == "gloo" and d["world"] == 2 and d["allreduce_checksum_ok"]
    assert d["optimizer_steps_in_timed_region"] >= 1 and d["allreduce_early_elements"] > 0
    assert 0 < line["allocator_priming_episodes"] <= 48 and line["allocator_priming_episodes"] % 8 == 0      # whole intervals of 16 / 2 episodes
    assert line["weights_finite"] and line["value"] > 0


def test_train_cli_synthetic_and_clip_directory(dev, tmp_path):
    """python -m litemkd_amd.train (reference flags, options.py:7-76): a few iterations on synthetic episodes with a checkpoint in the
    reference's format, then on a directory of decoded uint8 clips through the GPU frame transform (video_reader.py:398-485 sampling)"""
    import glob
    import json
    import numpy as np
    from litemkd_amd import train as T
    ck = tmp_path / "ckpt"
    log = tmp_path / "log.jsonl"
    losses, accs = T.main(["--shot", "1", "--query_per_class", "1", "--img_size", "64", "--tasks_per_batch", "2", "--training_iterations", "6",
                           "--print_freq", "2", "--checkpoint_dir", str(ck), "--learning_rate", "0.001", "--log_jsonl", str(log), "--test_iters", "100"])
    assert len(losses) == 6 and all(np.isfinite(losses))
    recs = [json.loads(ln) for ln in open(log)]
    assert recs and all("loss" in r and "lr" in r for r in recs)
    files = glob.glob(str(ck / "*.pt"))
    assert len(files) == 1
    sd = torch.load(files[0])["model_state_dict"]
    assert "backbone.resnet.4.0.conv1.weight" in sd and "classifier.transformers.k_linear.weight" in sd      # the reference's keys
    # a tiny dataset of decoded clips: 6 classes x 3 videos of 11 frames of 100 x 130, teacher features next to them
    root = tmp_path / "clips"
    g = np.random.default_rng(0)
    for c in range(6):
        d = root / ("class%02d" % c)
        d.mkdir(parents=True)
        for v in range(3):
            np.save(d / ("v%d.npy" % v), g.integers(0, 256, size=(11, 100, 130, 3), dtype=np.uint8))
            np.save(d / ("v%d.feature.npy" % v), g.standard_normal((8, 2048)).astype(np.float32))
    losses, accs = T.main(["--shot", "1", "--query_per_class", "1", "--img_size", "84", "--tasks_per_batch", "2", "--training_iterations", "4",
                           "--no_save", "--data_dir", str(root), "--dtype", "bf16", "--test_iters", "100", "--seed", "3"])
    assert len(losses) == 4 and all(np.isfinite(losses))


def test_rccl_one_rank_rehearsal(dev):
    """every RCCL call of the multi-GPU path on the real backend (process group "nccl", ONE rank - the test box has one GPU, and two ranks
    of one communicator cannot share it): the early all-reduce of the bucket's tail from its communication stream, the head's all-reduce,
    broadcast, the float64 MAX reductions, all_gather_object, barrier, the BatchNorm pooling - tools/rccl_rehearsal.py in a process of its
    own (a process group is per process).  The two-rank arithmetic is covered over gloo (test_data_parallel_world2_equals_world1)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_rehearsal.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL one-rank rehearsal ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
