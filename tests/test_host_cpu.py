"""CPU tests: C-ABI library loads and exports every symbol of include/lmkd.h, registry / state_dict
surface, loop semantics (with stand-in models), flat gradient bucket + gloo all-reduce (world_size 2)."""
import os
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    import litemkd_amd
    from litemkd_amd._lib import parse_header, LIB_PATH
    protos = parse_header()
    assert len(protos) >= 40
    L = litemkd_amd.lib()                      # binding fails if any declared symbol is missing
    assert L.value("lmkd_abi_version") == 3
    out = subprocess.check_output(["nm", "-D", LIB_PATH]).decode()
    for name in protos:
        assert (" T " + name) in out, name


def test_no_cpu_fallback():
    from litemkd_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear_fwd(torch.zeros(4, 8), torch.zeros(4, 8), None)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lite-mkd_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(d, f)


def test_registry_and_state_dict_keys():
    from litemkd_amd.model import model_select as ms
    from litemkd_amd.options import default_args
    from oracle import ref_cpu as O
    a = default_args()
    s = ms.Student(a)
    assert sum(p.numel() for p in s.parameters()) == 22721856
    sd = s.state_dict()
    for k, shp in O.resnet18_trunk_param_shapes().items():
        assert tuple(sd["backbone.resnet." + k].shape) == tuple(shp), k
    for k in ("backbone.fc1.weight", "backbone.fc2.bias", "classifier.transformers.k_linear.weight",
              "classifier.transformers.v_linear.bias", "classifier.transformers.norm_k.weight",
              "classifier.transformers.norm_v.bias", "classifier.transformers.pe.pe"):
        assert k in sd, k
    assert tuple(sd["classifier.transformers.pe.pe"].shape) == (1, 12, 2048)
    assert torch.allclose(sd["classifier.transformers.pe.pe"], O.positional_encoding_table(2048, 12))
    t = ms.Teacher(a)
    assert "classifier.transformers.k_linear.weight" in t.state_dict()
    with pytest.raises(KeyError):
        ms.select_model_student(default_args(model_backbone="nope"))
    with pytest.raises(KeyError):
        ms.select_model_teacher(default_args(model_teacher="nope"))
    with pytest.raises(NotImplementedError):
        ms.select_model_student(default_args(model_backbone="strmbackbone"))
    # DataParallel prefix stripping (model_select.py:143-150)
    out = ms.strip_dataparallel_prefix({"backbone.resnet.module.0.weight": 1, "backbone.fc1.weight": 2})
    assert set(out) == {"backbone.resnet.0.weight", "backbone.fc1.weight"}
    for name in ("resnet18_student", "resnet18_2fc", "resnet50_2fc", "resnet50_student"):
        assert ms.name2backbone[name] is not None
    from litemkd_amd.distillers import Distiller
    d = Distiller("fc_2_sup_dist", a.cfg, "cpu")
    assert callable(getattr(d, "fc_2_sup_dist")) and callable(getattr(d, "KD"))
    assert callable(getattr(d, "KL_feature"))        # every method of the reference's Distiller exists
    with pytest.raises(AttributeError):
        d.not_a_method


def test_loop_quirks_with_standins():
    """optimizer fires when (iteration+1) % 16 == 0 -> first after 15 episodes; scheduler every episode."""
    from litemkd_amd import trainloop as TL
    from litemkd_amd.options import default_args
    cfg = default_args(training_iterations=40, device=torch.device("cpu"), distill_name="fake", print_freq=10)
    w = torch.nn.Parameter(torch.zeros(1))
    events = []

    class Opt:
        def step(self):
            events.append(("step", it[0]))

        def zero_grad(self):
            pass

    class Sch:
        n = 0

        def step(self):
            Sch.n += 1

    it = [0]

    def student(ci, cl, ti):
        it[0] += 1
        return {"logits": (w * 0 + torch.zeros(5, 5))}

    def teacher(cf, cl, tf):
        return {"logits": torch.zeros(5, 5)}

    dist = types.SimpleNamespace(fake=lambda s, t, l: {"loss": s.sum()})
    src = TL.SyntheticEpisodes(default_args(shot=1, query_per_class=1, img_size=8), length=100)
    losses, accs = TL.train(student, teacher, src, dist, Opt(), Sch(), lambda lg, lb: torch.tensor(0.25), cfg)
    assert len(losses) == 40 and Sch.n == 40
    assert [e[1] for e in events] == [15, 31, 39]        # (it+1)%16==0 at 15, 31; final flush at total-1 = 39
    s = TL.MultiStepLR(types.SimpleNamespace(lr=1e-4), [3, 5])
    lrs = []
    for _ in range(6):
        s.step()
        lrs.append(s.opt.lr)
    assert lrs == pytest.approx([1e-4, 1e-4, 1e-5, 1e-5, 1e-6, 1e-6])


def test_synthetic_episode_contract():
    from litemkd_amd import trainloop as TL
    from litemkd_amd.options import default_args
    ep = TL.SyntheticEpisodes(default_args(shot=1, img_size=16), base_seed=3).episode(0)
    assert ep["support_set"].shape == (1, 40, 3, 16, 16) and ep["target_set"].shape == (1, 200, 3, 16, 16)
    assert ep["support_set_feature_teacher"].shape == (1, 5, 8, 2048)
    assert ep["support_labels"].dtype == torch.float32 and sorted(ep["target_labels"][0].tolist()) == sorted([0., 1, 2, 3, 4] * 5)
    assert 0 <= float(ep["support_set"].min()) and float(ep["support_set"].max()) < 1
    out = TL.prepare_task(ep, torch.device("cpu"))
    assert out[5].dtype == torch.int64 and out[0].shape == (40, 3, 16, 16)


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from litemkd_amd.parallel import FlatParams, init_distributed
    init_distributed("gloo")
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    fp = FlatParams(m)
    fp.broadcast_params(0)
    x = torch.full((4, 5), float(rank + 1))
    m(x).sum().backward()
    m(x).sum().backward()                         # accumulation into the flat views
    local = fp.grad.clone()
    fp.allreduce_grads()
    # checkpoint-time BatchNorm averaging (one collective) and the scheduler's global-episode count
    from litemkd_amd.parallel import sync_bn_running_stats
    from litemkd_amd.trainloop import MultiStepLR
    bn = torch.nn.Sequential(torch.nn.BatchNorm1d(3), torch.nn.BatchNorm1d(2))
    with torch.no_grad():
        bn[0].running_mean.fill_(float(rank + 1))
        bn[1].running_var.fill_(10.0 * (rank + 1))
    avg = sync_bn_running_stats(bn)

    class _O:
        lr = 1.0
    sch = MultiStepLR(_O(), [4])
    sch.step()
    lr1 = sch.opt.lr
    sch.step()
    # numpy copies, pickled by value: a torch tensor on an mp queue travels as a shared-memory file that the parent must open before this
    # process exits (it failed with FileNotFoundError once in a while)
    q.put((rank, local.numpy().copy(), fp.grad.numpy().copy(), [p.grad.data_ptr() for p in m.parameters()], fp.grad.data_ptr(),
           {k: v.numpy().copy() for k, v in avg.items()}, bn[0].running_mean.numpy().copy(), (lr1, sch.opt.lr)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(60)
    T = torch.from_numpy
    res = [(r, T(l), T(s), p, b, {k: T(v) for k, v in a.items()}, T(lv), lr) for r, l, s, p, b, a, lv, lr in res]
    (r0, l0, s0, ptrs0, base0, avg0, live0, lrs0), (r1, l1, s1, _, _, avg1, live1, _) = res
    assert torch.allclose(s0, l0 + l1) and torch.allclose(s1, l0 + l1)
    assert ptrs0[0] == base0                       # .grad tensors are views into the bucket
    assert float(l0.abs().sum()) > 0 and not torch.allclose(l0, l1)
    # BatchNorm running statistics averaged over the ranks for the checkpoint; the live buffers stay per-rank
    assert set(avg0) == {"0.running_mean", "0.running_var", "1.running_mean", "1.running_var"}
    assert torch.allclose(avg0["0.running_mean"], torch.full((3,), 1.5)) and torch.allclose(avg1["1.running_var"], torch.full((2,), 15.0))
    # pooled variance = mean of the ranks' variances (1, 1) + the spread of their means (1, 2): 1 + (2.5 - 1.5^2)
    assert torch.allclose(avg0["0.running_var"], torch.full((3,), 1.25))
    assert float(live0[0]) == 1.0 and float(live1[0]) == 2.0
    # MultiStepLR counts GLOBAL episodes: at world 2 the milestone 4 is reached after 2 local episodes
    assert lrs0 == (1.0, 0.1) or abs(lrs0[1] - 0.1) < 1e-12 and lrs0[0] == 1.0


def test_resize_plan_tables_match_pillow_restatement():
    """lmkd_resize_plan (host-side C, no GPU needed): bounds and 22-bit coefficients of Pillow's BILINEAR resampler"""
    import ctypes
    import numpy as np
    import litemkd_amd
    from oracle import ref_cpu as O
    L = litemkd_amd.lib()
    for n_in, n_out in [(320, 341), (240, 256), (1280, 455), (131, 50), (48, 256), (7, 3), (5, 5)]:
        b_ref, k_ref, ks_ref = O.pil_bilinear_coeffs(n_in, n_out)
        ks = L.value("lmkd_resize_plan", n_in, n_out, None, None)
        assert ks == ks_ref
        b = np.zeros((n_out, 2), np.int32)
        k = np.zeros((n_out, ks), np.int32)
        assert L.value("lmkd_resize_plan", n_in, n_out, ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(k.ctypes.data)) == ks
        assert np.array_equal(b, b_ref) and np.array_equal(k, k_ref)


def test_resized_shape_and_identity_plan():
    """host logic of the one-launch frame transform (ops.frames_resize_crop_nhwc4, round 5): the resized shape equals the oracle's restatement of
    functional.py:44-59 for int and (h, w) sizes, and a plan between equal sizes is the exact identity (one tap of weight 2^22), so the kernel
    needs no special case where Resize leaves an axis alone; bad arguments of the entry point come back as error codes before any launch."""
    import ctypes
    import numpy as np
    import litemkd_amd
    from litemkd_amd import ops
    from oracle import ref_cpu as O
    for (h, w) in [(240, 320), (320, 240), (256, 300), (256, 256), (288, 352), (100, 256), (17, 9)]:
        for size in (256, 128, 17):
            assert ops._resized_shape(h, w, size) == O.resize_short_side(h, w, size), (h, w, size)
    assert ops._resized_shape(240, 320, (80, 72)) == (80, 72)
    L = litemkd_amd.lib()
    for n in (5, 224, 341):
        ks = L.value("lmkd_resize_plan", n, n, None, None)
        b = np.zeros((n, 2), np.int32)
        k = np.zeros((n, ks), np.int32)
        L.value("lmkd_resize_plan", n, n, ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(k.ctypes.data))
        assert np.array_equal(b[:, 0], np.arange(n)) and int(k[:, 0].min()) == 1 << 22 and int(np.abs(k[:, 1:]).max()) == 0
    rc = L.cdll.lmkd_frames_resize_crop_nhwc4(None, None, None, None, 3, None, None, 3, None, None, None, 8, 240, 320, 256, 341, 224, 224, 8, None)
    assert rc != 0 and b"lmkd_frames_resize_crop_nhwc4" in L.cdll.lmkd_last_error()


def test_cabi_argument_errors_return_codes_not_crashes():
    """The C ABI never throws and never launches on bad arguments: negative return code + lmkd_last_error() text, and the
    Python shim turns that into RuntimeError (SURVEY.md 8b 'errors').  No GPU is touched: every check precedes the launch."""
    import ctypes
    import pytest
    import litemkd_amd
    L = litemkd_amd.lib()
    cd = L.cdll
    cd.lmkd_last_error.restype = ctypes.c_char_p
    null = ctypes.c_void_p(0)
    # raw C calls
    assert cd.lmkd_conv2d_fwd(null, null, null, null, 1, 8, 8, 32, 64, 3, 3, 1, 1, null) < 0
    assert b"null pointer" in cd.lmkd_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert cd.lmkd_conv2d_fwd(p, p, p, null, 1, 8, 8, 7, 64, 3, 3, 1, 1, null) < 0          # Cs = 7: not 4 or a multiple of 32
    assert b"channel count 7" in cd.lmkd_last_error()
    assert cd.lmkd_conv_set_tile(99) < 0 and b"lmkd_conv_set_tile" in cd.lmkd_last_error()
    assert cd.lmkd_conv_set_compute_dtype(17) < 0
    assert cd.lmkd_resize_plan(0, 10, null, null) < 0
    assert cd.lmkd_trx_sup_sim_fwd(p, p, p, p, 4, 9, 3, 28, 1152, null) < 0                  # way > 8
    assert cd.lmkd_conv2d_split_weights(p, p, 48, 64, null) < 0                              # ncols not a multiple of 32
    assert cd.lmkd_bn_apply(p, p, null, null, p, 4, 6, 0, 0, null, null) < 0                       # C % 4 != 0
    # the shim
    with pytest.raises(RuntimeError, match="lmkd_conv_set_tile"):
        L.call("lmkd_conv_set_tile", 99)
    assert cd.lmkd_conv_set_tile(0) == 0 and cd.lmkd_conv_set_compute_dtype(2) == 0


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
    child process (never an exec of a process that touched the GPU) and returns its exit code"""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "8", "--warmup", "2"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "8", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert len(bench.kernel_source_hash()) == 16


def test_make_returns_the_reference_tuple():
    """trainloop.make mirrors trainwandb.py:78-109: (student, teacher, video_loader, distillers, accuracy_fn, test_accuracies,
    optimizer, scheduler) from the reference's args namespace; runs on the CPU (construction only, no kernel is launched)."""
    import litemkd_amd  # noqa: F401
    from litemkd_amd import trainloop as TL
    from litemkd_amd.options import default_args
    cfg = default_args(device=torch.device("cpu"), shot=1, query_per_class=1, img_size=32, opt="adam", learning_rate=3e-4, sch=[7, 9])
    student, teacher, loader, distillers, accuracy_fn, test_acc, optimizer, scheduler = TL.make(cfg, base_seed=3)
    assert type(student).__name__ == "Student" and type(teacher).__name__ == "Teacher"
    assert optimizer.opt == "adam" and optimizer.lr == 3e-4 and scheduler.milestones == [7, 9] and test_acc.datasets == ["hmdb"]
    assert callable(accuracy_fn) and hasattr(distillers, cfg.distill_name)
    ep = next(iter(loader))
    assert ep["support_set"].shape == (1, 40, 3, 32, 32) and ep["support_labels"].shape == (1, 5)
    # every trainable parameter is a view into the optimizer's flat buffer, gradients zeroed
    p0 = next(student.parameters())
    assert p0.data_ptr() == optimizer.bucket.flat.data_ptr() and float(optimizer.bucket.grad.abs().max()) == 0.0
    with pytest.raises(KeyError):
        TL.make(default_args(device=torch.device("cpu"), opt="rmsprop"))


# the command line of the reference's train_wandb.sh (its exported variables substituted; a literal copy of the FLAG NAMES AND VALUES the
# script passes to trainwandb.py - data, not code)
_TRAIN_WANDB_SH = ("--dataset hmdb --shot 5 --debug False --num_gpus 1 --mode hmdb --test_model student --num_test_tasks 10000 "
                   "--learning_rate 0.0001 --checkpoint_dir hmdb_ckpt/ --training_iterations 70010 --model_backbone resnet18_2fc "
                   "--model_classifier TRX_2fcsup --model_teacher test_teacher_TRX_2fcsup_fixed --teacher_checkpoint /data/hmdb/teacher/checkpoint35000.pt "
                   "--distill_name fc_2_sup_dist --temp_set 2 --trans_linear_in_dim 2048")


def test_train_cli_parses_the_reference_command_line():
    """python -m litemkd_amd.train takes the flags of options.py:7-76 under their own names: train_wandb.sh's command line parses
    unchanged, and every option the reference's parser declares exists here with the reference's default (plugin names excepted)"""
    from litemkd_amd import train as T
    p = T.build_parser()
    a = p.parse_args(_TRAIN_WANDB_SH.split())
    assert (a.dataset, a.shot, a.debug, a.num_gpus, a.mode, a.test_model, a.num_test_tasks) == ("hmdb", 5, False, 1, "hmdb", "student", 10000)
    assert (a.learning_rate, a.training_iterations, a.model_backbone, a.model_classifier) == (1e-4, 70010, "resnet18_2fc", "TRX_2fcsup")
    assert (a.model_teacher, a.distill_name, a.temp_set, a.trans_linear_in_dim) == ("test_teacher_TRX_2fcsup_fixed", "fc_2_sup_dist", [2], 2048)
    d = p.parse_args([])
    ref_defaults = dict(way=5, shot=5, query_per_class=5, query_per_class_test=1, tasks_per_batch=16, print_freq=10, seq_len=8, num_workers=1,
                        trans_linear_out_dim=1152, trans_linear_in_dim=2048, img_size=224, temp_set=[2], trans_dropout=0.1, save_freq=10000,
                        split=3, sch=[20000, 40000], num_test_tasks=5000, method="resnet18", num_gpus=1, dataset="kinetics", mode="KD_KL_meta",
                        debug=False, soft_loss_weight=1, hard_loss_weight=1, test=False, checkpoint_dir=None, training_iterations=100010,
                        resume_from_checkpoint=False, learning_rate=0.0001, opt="sgd",
                        test_iters=[10000, 15000, 20000, 30000, 35000, 40000, 50000, 60000, 70000, 80000, 90000, 100000])
    for k, v in ref_defaults.items():
        assert getattr(d, k) == v, k
    assert d.cfg["temperature"] == 4 and d.cfg["soft_loss_weight"] == 2
    # the default arithmetic and schedule are the ones bench.py times; the reference's fp32 semantics under other arithmetics stay selectable
    s = T.schedule_from_args(d)
    assert (s.conv_dtype, s.act_dtype, s.merge_trunk_calls, s.pipeline_episodes) == ("fp32h2", "fp32", True, True)
    assert T.schedule_from_args(p.parse_args(["--dtype", "f32x3"])).conv_dtype == "fp32x3"
    assert T.schedule_from_args(p.parse_args(["--dtype", "bf16"])).act_dtype == "bf16"
    assert not T.schedule_from_args(p.parse_args(["--dtype", "f32native"])).merge_trunk_calls
    # when the reference tree is at hand (the build container), every `--flag` its parser and its script name is known here
    ref = "/root/reference"
    if os.path.isdir(ref):
        import re
        known = set(p._option_string_actions)
        for f in ("options.py", "train_wandb.sh"):
            txt = open(os.path.join(ref, f)).read()
            if f == "options.py":      # the training parser: parse_common_args + parse_train_args
                txt = txt[:txt.index("def parse_test_args")]
            for flag in re.findall(r"(--[a-z_]+)", txt):
                assert flag in known, (f, flag)
    with pytest.raises(SystemExit):
        T.args_check(p.parse_args([]))          # "need to specify a checkpoint dir" (options.py:89-91)


def test_schedule_object_on_cpu():
    from litemkd_amd.schedule import Schedule
    s = Schedule.from_env({"LMKD_MERGE": "1", "LMKD_SIDE_WGRAD": "0"})
    assert s.merge_trunk_calls and not s.side_wgrad and s.overlap_trunk_calls
    ser = Schedule.serial()
    assert not ser.overlap_trunk_calls and not ser.direct_param_grad and ser.sync_wgrad_at_backward_end
    before = Schedule.current()
    ser.apply()
    try:
        from litemkd_amd import ops, trainloop
        from litemkd_amd.model.backbone import resnet
        assert not ops.SIDE_WGRAD and not resnet.OVERLAP_TRUNK_CALLS and not trainloop.TEACHER_STREAM
        assert Schedule.current() == ser
    finally:
        before.apply()
