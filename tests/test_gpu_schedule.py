"""GPU: the schedule the benchmark TIMES against the serial schedule, at the benchmark's size.

bench.py runs the episode loop under schedule.Schedule.bench(): both trunk calls as one launch per layer, the forward of episode i + 1
queued beside the backward of episode i (two stream sets), weight gradients on a third stream, the frozen teacher head and the second
TRX head on auxiliary streams, BatchNorm / Linear / TRX parameter gradients added into .grad (or its per-stream shadow) by the
kernels, all weight packs re-packed in place at the optimizer step (round 3's schedule - two trunk calls on two streams, no
pipelining - is Schedule.two_call()).  The oracle tests run the SERIAL schedule
(one stream, every gradient through autograd).  A missing event wait between streams is size dependent - at 64 px a kernel lasts
microseconds and nothing can overtake it - so the identity of the two schedules is checked HERE at 400 frames of 224^2
(trainwandb.py:122-143: two accumulated episodes, then optimizer.step()), and in the small cases that bend the dependency structure
(a frozen stem weight: ADVICE round 3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import litemkd_amd
    litemkd_amd.lib().call("lmkd_device_check", 0)
    return torch.device("cuda", 0)


def _run(dev, sched, shot, img, episodes, freeze=(), opt="sgd", lr=1e-2, seed=3):
    """`episodes` accumulated training episodes + one optimizer step under `sched` -> (losses, flat gradient bucket before the step,
    flat weights after it, running statistics)"""
    from litemkd_amd import ops, trainloop as TL
    from litemkd_amd.options import default_args
    cfg = default_args(shot=shot, img_size=img, trans_dropout=0.1, device=dev, opt=opt, learning_rate=lr)
    with sched.applied():
        torch.manual_seed(seed)
        student, teacher = TL.init_model(cfg)
        for n, p in student.named_parameters():
            if n in freeze:
                p.requires_grad_(False)
        from litemkd_amd.distillers import Distiller
        from litemkd_amd.utils import aggregate_accuracy
        distiller = Distiller(cfg.distill_name, cfg.cfg, dev)
        optim = TL.FusedOptimizer(student, opt, lr)
        optim.zero_grad()
        src = TL.SyntheticEpisodes(cfg, base_seed=77, device=dev)
        pool = [src.episode(e) for e in range(episodes)]
        torch.manual_seed(seed + 1)                      # the TRX dropout seeds are drawn from torch's generator, in program order
        losses = []
        if sched.pipeline_episodes:                      # the forward of episode i + 1 is queued before the backward of episode i (bench.py's loop)
            pipe = TL.PipelinedEpisodes(student, teacher, distiller, aggregate_accuracy, cfg)
            for ep in pool:
                r = pipe.push(ep)
                if r is not None:
                    losses.append(r[0])
            r = pipe.flush()
            losses += [x[0] for x in (r if isinstance(r, list) else [r])]
        else:
            for ep in pool:
                loss, _, _ = TL.train_task(ep, student, teacher, distiller, aggregate_accuracy, cfg)
                losses.append(loss)
        ops.join_all_streams()
        optim.bucket.fold_shadow()
        grad = optim.bucket.grad.clone()
        bufs = {n: b.clone() for n, b in student.named_buffers() if "running" in n}      # (before the probe forward below updates them again)
        optim.step()
        optim.zero_grad()
        # one more forward on the re-packed weights: a stale pack (REPACK_AT_STEP / the pack cache) would show here
        with torch.no_grad():
            pr = TL.prepare_task(pool[0], dev)
            probe = student(pr[0], pr[4], pr[1])["logits"]["kl"].clone()
        torch.cuda.synchronize()
        flat = optim.bucket.flat.clone()
        return torch.stack(losses).cpu(), grad, flat, bufs, probe


def _compare(a, b, grad_tol, lr):
    la, ga, wa, ba, pa = a
    lb, gb, wb, bb, pb = b
    assert torch.equal(la, lb), (la, lb)                                  # the forward is the same arithmetic in the same order
    gmax = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= grad_tol * gmax, (float((ga - gb).abs().max()) / gmax)
    # SGD: w' = w - lr g, so the weights may differ by lr x the gradient difference (+ one rounding of the weight itself)
    assert float((wa - wb).abs().max()) <= lr * grad_tol * gmax + 1.2e-7 * float(wb.abs().max()), float((wa - wb).abs().max())
    for n in bb:
        assert torch.equal(ba[n], bb[n]), n                               # running statistics: support call, then query call
    assert float((pa - pb).abs().max()) <= 1e-4 * float(pb.abs().max()), "forward after the step"


def _sched(which, conv="fp32x3"):
    from litemkd_amd.schedule import Schedule
    return {"bench": lambda: Schedule.bench(conv_dtype=conv), "two_call": lambda: Schedule.two_call(conv_dtype=conv),
            "merged": lambda: Schedule.bench(pipeline_episodes=False, conv_dtype=conv),
            "two_call_pipelined": lambda: Schedule.two_call(pipeline_episodes=True, conv_dtype=conv)}[which]()


@pytest.mark.parametrize("which,conv", [("bench", "fp32h2"), ("bench", "fp32x3"), ("two_call", "fp32x3"), ("merged", "fp32x3")])
def test_bench_schedule_equals_serial_full_size(dev, which, conv):
    """three accumulated 400-frame episodes (5-way 5-shot, 224^2) + one SGD step under Schedule.bench() - what bench.py times: merged trunk
    call + cross-episode pipelining - under round 3's two-call three-stream schedule and under the merged call alone, against
    Schedule.serial(): losses identical, flat gradient bucket within 2e-6 of its maximum, weights after the step equal to that precision,
    running statistics identical"""
    from litemkd_amd.schedule import Schedule
    import litemkd_amd
    assert Schedule.bench().merge_trunk_calls and Schedule.bench().pipeline_episodes
    n0 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    ref = _run(dev, Schedule.serial(conv_dtype=conv), 5, 224, 3)
    n1 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    tst = _run(dev, _sched(which, conv), 5, 224, 3)
    n2 = litemkd_amd.lib().value("lmkd_conv_h2_launches")
    if conv == "fp32h2":      # bench.py's headline arithmetic: the two-plane kernels ran - 56 launches per trunk call (1 + 16 forward, 16 + 3 data
        # gradient, 1 + 16 + 3 weight gradient; the 1s: the stem) x 2 calls x 3 episodes in the serial run, once per episode in the merged
        # one, + the 17 forward launches per call of _run's probe - i.e. every maximum reached its consumer
        assert n1 - n0 == 2 * (n2 - n1) and n2 - n1 == 3 * 56 + 17, (n1 - n0, n2 - n1)
    else:
        assert n2 == n0
    _compare(tst, ref, 2e-6, 1e-2)


@pytest.mark.parametrize("which,conv", [("bench", "fp32x3"), ("two_call", "fp32x3"), ("two_call_pipelined", "fp32x3"), ("bench", "fp32h2")])
@pytest.mark.parametrize("freeze", [(), ("backbone.resnet.0.weight",), ("backbone.resnet.0.weight", "backbone.resnet.1.weight", "backbone.resnet.1.bias")])
def test_bench_schedule_equals_serial_frozen_stem(dev, which, conv, freeze):
    """the same identity on a small episode with the stem's weight (and BatchNorm) frozen: the stem's weight gradient is then NOT the
    last thing on the side stream that the weight-gradient stream waits for - the optimizer has to join every stream itself
    (ops.join_all_streams)"""
    from litemkd_amd.schedule import Schedule
    ref = _run(dev, Schedule.serial(conv_dtype=conv), 1, 64, 4, freeze)
    tst = _run(dev, _sched(which, conv), 1, 64, 4, freeze)
    _compare(tst, ref, 5e-6, 1e-2)


def test_schedule_object_round_trip(dev):
    from litemkd_amd.schedule import Schedule
    from litemkd_amd import ops
    before = Schedule.current()
    with Schedule.serial(conv_dtype="bf16", act_dtype="bf16").applied() as s:
        assert not ops.SIDE_WGRAD and ops.SYNC_WGRAD_AT_BACKWARD_END and ops.get_conv_compute_dtype() == "bf16" and ops.get_activation_dtype() == "bf16"
        assert Schedule.current() == s
    assert Schedule.current() == before
