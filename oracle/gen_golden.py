"""Generate tests/golden/*.npz by running the REFERENCE's own modules in the build
container, and check the oracle restatement (oracle/ref_cpu.py) against them.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (the build container).
The fixtures it writes are data only (seeded inputs + the reference's outputs); no
reference source text or bytecode is stored.

Reference modules imported (by file path):
  distillers.py, utils.py                          -- import as-is (torch only)
  model/classifiers/e_dist_fc2.py                  -- import as-is
  model/classifiers/TRX_2fcsup.py                  -- needs two test-only shims: an empty
      `torchvision.models` stub for a dead import (TRX_2fcsup.py:12) and
      `Tensor.cuda = identity` for the tuple-index tensors (TRX_2fcsup.py:71)

  model/backbone/resnet18_2fc.py, model/model_select.py (A3 pooled head + fc1/fc2, A1 Student.forward dict plumbing)
      -- `torchvision.models.resnet18` is a test-only stub whose children()[:-2] is an identity, so the tensors fed in ARE the
      trunk's output maps; every line after `self.resnet(...)` is the reference's own (gen_head)

  teacher/code/model.py (MFM fusion)                -- test-only stubs for dead imports: `turtle`, `timm`,
      `torchvision.models` (matplotlib, einops and the local `transformer` / `utils` modules import as they are);
      `Tensor.cuda = identity` for extract_feature's `.cuda()` calls (model.py:1649-1651)

usage: python oracle/gen_golden.py [--check-only]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_reference():
    sys.path.insert(0, REF)
    mods = {}
    mods["distillers"] = _load("ref_distillers", os.path.join(REF, "distillers.py"))
    mods["utils"] = _load("utils", os.path.join(REF, "utils.py"))       # TRX imports `utils`
    sys.modules["utils"] = mods["utils"]
    mods["e_dist_fc2"] = _load("ref_e_dist_fc2", os.path.join(REF, "model/classifiers/e_dist_fc2.py"))
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.models = types.ModuleType("torchvision.models")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv.models
    torch.Tensor.cuda = lambda self, *a, **k: self
    mods["trx"] = _load("ref_TRX_2fcsup", os.path.join(REF, "model/classifiers/TRX_2fcsup.py"))
    mods["trx_sup"] = _load("ref_TRX_sup", os.path.join(REF, "model/classifiers/TRX_sup.py"))     # same two shims
    return mods


class Args:
    way = 5
    shot = 5
    seq_len = 8
    trans_linear_out_dim = 1152
    trans_linear_in_dim = 2048
    trans_dropout = 0.1
    query_per_class = 5


def gsum(g):
    """compact digest of a [N,8,2048] gradient: 64-wide chunk sums (keeps fixtures small)."""
    return g.reshape(g.shape[0], 8, 32, 64).sum(-1)


def t2n(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_distill(mods):
    """kd_loss / inter_class_relation / fc_2_sup_dist / KD / Dist_KD values + input grads."""
    D = mods["distillers"]
    out = {}
    for case, (nq, seed) in enumerate([(25, 11), (25, 12), (5, 13)]):
        g = torch.Generator().manual_seed(seed)
        s = {"kl": (torch.randn(nq, 5, generator=g) * 30 - 300).requires_grad_(),
             "ce": (torch.randn(nq, 5, generator=g) * 30 - 300).requires_grad_(),
             "sup": (torch.randn(5, 4, generator=g) * 20 - 100).requires_grad_()}
        t = {"kl": torch.randn(nq, 5, generator=g) * 30 - 300,
             "sup": torch.randn(5, 4, generator=g) * 20 - 100}
        labels = torch.randint(0, 5, (nq,), generator=g)
        dist = D.Distiller("fc_2_sup_dist", O.DEFAULT_CFG, torch.device("cpu"))
        r = dist.fc_2_sup_dist(s, t, labels)
        r["loss"].backward()
        pre = "c%d_" % case
        out.update({pre + "s_kl": s["kl"], pre + "s_ce": s["ce"], pre + "s_sup": s["sup"],
                    pre + "t_kl": t["kl"], pre + "t_sup": t["sup"], pre + "labels": labels,
                    pre + "loss": r["loss"], pre + "soft": r["soft_loss"], pre + "hard": r["hard_loss"],
                    pre + "g_kl": s["kl"].grad, pre + "g_ce": s["ce"].grad, pre + "g_sup": s["sup"].grad,
                    pre + "kd": D.kd_loss(s["kl"], t["kl"], 4), pre + "icr": D.inter_class_relation(s["sup"], t["sup"])})
        # single-logit methods
        s1 = s["kl"].detach().clone().requires_grad_()
        r = dist.KD(s1, t["kl"], labels)
        r["loss"].backward()
        out.update({pre + "KD_loss": r["loss"], pre + "KD_g": s1.grad})
        s2 = s["kl"].detach().clone().requires_grad_()
        r = dist.Dist_KD(s2, t["kl"], labels)
        r["loss"].backward()
        out.update({pre + "DistKD_loss": r["loss"], pre + "DistKD_g": s2.grad})
        acc = mods["utils"].aggregate_accuracy(s["kl"].detach() + s["ce"].detach(), labels)
        out.update({pre + "acc": acc, pre + "argmax": torch.argmax(s["kl"].detach() + s["ce"].detach(), -1)})
    return out


def gen_distill_methods(mods):
    """every logits-only Distiller method of the reference: loss value + gradient of every student logit tensor"""
    D = mods["distillers"]
    out = {}
    for i, name in enumerate(sorted(O.DISTILL_SIGNATURES)):
        s, t, labels = O.distill_inputs(name, 500 + i)
        leaves = [s] if torch.is_tensor(s) else list(s.values())
        for v in leaves:
            v.requires_grad_()
        dist = D.Distiller(name, dict(O.DEFAULT_CFG), torch.device("cpu"))
        r = getattr(dist, name)(s, t, labels)
        r["loss"].sum().backward()
        out[name + "__loss"] = r["loss"].detach().reshape(())
        if torch.is_tensor(s):
            out[name + "__g"] = s.grad
        else:
            for k, v in s.items():
                out[name + "__g_" + k] = v.grad if v.grad is not None else torch.zeros_like(v)
    return out


def kl_feature_inputs(seed, nq=25, nv=50):
    g = torch.Generator().manual_seed(seed)
    s = {"logits": torch.randn(nq, 5, generator=g) * 30 - 300, "feature": torch.randn(nv, 8, 2048, generator=g)}
    t = {"logits": torch.randn(nq, 5, generator=g) * 30 - 300, "feature": torch.randn(nv, 8, 2048, generator=g) * 0.8 + 0.1}
    return s, t, torch.randint(0, 5, (nq,), generator=g)


def gen_kl_feature(mods):
    """Distiller.KL_feature (distillers.py:126-150) on the dicts train_task builds (trainwandb.py:209-226)"""
    D = mods["distillers"]
    out = {}
    for case, (seed, nq, nv) in enumerate([(61, 25, 50), (62, 5, 10)]):
        s, t, labels = kl_feature_inputs(seed, nq, nv)
        s["logits"].requires_grad_()
        s["feature"].requires_grad_()
        r = D.Distiller("KL_feature", dict(O.DEFAULT_CFG), torch.device("cpu")).KL_feature(s, t, labels)
        r["loss"].backward()
        pre = "c%d_" % case
        out.update({pre + "seed": seed, pre + "nq": nq, pre + "nv": nv, pre + "loss": r["loss"], pre + "hard": r["hard_loss"],
                    pre + "soft": r["soft_loss"], pre + "feature_loss": r["feature_loss"], pre + "g_logits": s["logits"].grad,
                    pre + "g_feature": gsum(s["feature"].grad), pre + "g_feature_row0": s["feature"].grad[0, :, :128].clone()})
    return out


def feature_case(seed, ns=25, nq=25, scale=1.0, shuffle=True):
    g = torch.Generator().manual_seed(seed)
    sup = torch.randn(ns, 8, 2048, generator=g) * scale
    qry = torch.randn(nq, 8, 2048, generator=g) * scale
    lab = torch.arange(5).repeat_interleave(ns // 5)
    if shuffle:
        lab = lab[torch.randperm(ns, generator=g)]
    return sup, lab.float(), qry


def gen_edist(mods):
    E = mods["e_dist_fc2"]
    args = Args()
    out = {}
    for case, (seed, ns, nq, shuffle) in enumerate([(21, 25, 25, True), (22, 25, 5, False), (23, 5, 25, True)]):
        args.shot = ns // 5
        sup, lab, qry = feature_case(seed, ns, nq, 1.0, shuffle)
        sup.requires_grad_()
        qry.requires_grad_()
        clf = E.e_dist_1fc_sup(args)
        r = clf(sup, lab, qry)["logits"]
        w_kl = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        w_sup = torch.linspace(1, -1, 20).reshape(5, 4)
        ((r["kl"] * w_kl).sum() + (r["sup"] * w_sup).sum() * 1e-3).backward()
        pre = "c%d_" % case
        out.update({pre + "seed": seed, pre + "ns": ns, pre + "nq": nq, pre + "shuffle": int(shuffle),
                    pre + "kl": r["kl"], pre + "sup": r["sup"],
                    pre + "g_sup_feat": gsum(sup.grad), pre + "g_qry_feat": gsum(qry.grad),
                    pre + "g_sup_row0": sup.grad[0, :, :128].clone(), pre + "g_qry_row0": qry.grad[0, :, :128].clone()})
    return out


def gen_trx(mods):
    """TRX_2fcsup / TRX_2fcsup_fixed in eval() (dropout off), weights = oracle.make_trx_params(seed)."""
    T = mods["trx"]
    out = {}
    for case, (seed, ns, nq, shuffle) in enumerate([(31, 25, 25, True), (32, 5, 25, True), (33, 25, 5, False)]):
        args = Args()
        args.shot = ns // 5
        g = torch.Generator().manual_seed(1000 + seed)
        p = O.make_trx_params(g)
        # non-trivial LayerNorm affine so the test sees it
        p["norm_k.weight"] = 1 + 0.1 * torch.randn(1152, generator=g)
        p["norm_k.bias"] = 0.1 * torch.randn(1152, generator=g)
        clf = T.TRX_2fcsup(args).eval()
        sd = clf.state_dict()
        for k, v in p.items():
            sd["transformers." + k].copy_(v)
        sup1, lab, qry1 = feature_case(seed, ns, nq, 0.5, shuffle)
        sup2, _, qry2 = feature_case(seed + 100, ns, nq, 0.5, shuffle)
        for x in (sup1, qry1, sup2, qry2):
            x.requires_grad_()
        ctx = {"context_features_1": sup1, "context_features_2": sup2}
        tgt = {"target_features_1": qry1, "target_features_2": qry2}
        r = clf(ctx, lab, tgt)["logits"]
        w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        w_sup = torch.linspace(1, -1, 20).reshape(5, 4)
        ((r["kl"] * w).sum() * 1e-2 + (r["ce"] * w.flip(0)).sum() * 1e-2 + (r["sup"] * w_sup).sum() * 1e-3).backward()
        pre = "c%d_" % case
        tr = clf.transformers
        out.update({pre + "seed": seed, pre + "ns": ns, pre + "nq": nq, pre + "shuffle": int(shuffle),
                    pre + "kl": r["kl"], pre + "ce": r["ce"], pre + "sup": r["sup"],
                    pre + "g_sup1": gsum(sup1.grad), pre + "g_qry1": gsum(qry1.grad),
                    pre + "g_sup2": gsum(sup2.grad), pre + "g_qry2": gsum(qry2.grad),
                    pre + "g_sup1_row0": sup1.grad[0, :, :128].clone(), pre + "g_qry1_row0": qry1.grad[0, :, :128].clone(),
                    pre + "g_kw_sum": tr.k_linear.weight.grad.sum(dim=1), pre + "g_kb": tr.k_linear.bias.grad,
                    pre + "g_vw_sum": tr.v_linear.weight.grad.sum(dim=1), pre + "g_vb": tr.v_linear.bias.grad,
                    pre + "g_nkw": tr.norm_k.weight.grad, pre + "g_nkb": tr.norm_k.bias.grad})
        fx = T.TRX_2fcsup_fixed(args).eval()
        fsd = fx.state_dict()
        for k, v in p.items():
            fsd["transformers." + k].copy_(v)
        rf = fx(sup1.detach(), lab, qry1.detach())["logits"]
        out.update({pre + "fixed_kl": rf["kl"], pre + "fixed_sup": rf["sup"]})
    return out


def gen_trx_sup(mods):
    """TRX_sup / TRX_sup_fixed (TRX_sup.py) in eval(), 20 queries (Distiller.support_sim hard-codes reshape(20,25)), and
    Distiller.support_sim on their outputs."""
    T = mods["trx_sup"]
    D = mods["distillers"]
    out = {}
    for case, (seed, ns, nq, shuffle) in enumerate([(41, 25, 20, True), (42, 5, 20, True)]):
        args = Args()
        args.shot = ns // 5
        p, sup, qry, sup_t, qry_t, lab = trx_case_inputs(seed, ns, nq, shuffle)
        clf = T.TRX_sup(args).eval()
        sd = clf.state_dict()
        for k, v in p.items():
            sd["transformers." + k].copy_(v)
        sup.requires_grad_()
        qry.requires_grad_()
        r = clf(sup, lab, qry)["logits"]
        fx = T.TRX_sup_fixed(args).eval()
        fsd = fx.state_dict()
        for k, v in p.items():
            fsd["transformers." + k].copy_(0.9 * v)           # a different (frozen) teacher
        rt = fx(sup_t, lab, qry_t)["logits"]
        labels = torch.arange(nq) % 5
        res = D.Distiller("support_sim", dict(O.DEFAULT_CFG), torch.device("cpu")).support_sim(r, rt, labels)
        w = torch.linspace(-1, 1, nq * 25).reshape(nq, 5, 5)
        (res["loss"] + (r["support_set"] * w).sum() * 1e-2).backward()
        pre = "c%d_" % case
        tr = clf.transformers
        out.update({pre + "seed": seed, pre + "ns": ns, pre + "nq": nq, pre + "shuffle": int(shuffle),
                    pre + "query": r["query"], pre + "support_set": r["support_set"],
                    pre + "fixed_query": rt["query"], pre + "fixed_support_set": rt["support_set"],
                    pre + "loss": res["loss"], pre + "hard_loss": res["hard_loss"], pre + "soft_support_loss": res["soft_support_loss"],
                    pre + "soft_query_loss": res["soft_query_loss"],
                    pre + "g_sup": gsum(sup.grad), pre + "g_qry": gsum(qry.grad),
                    pre + "g_kw_sum": tr.k_linear.weight.grad.sum(dim=1), pre + "g_vw_sum": tr.v_linear.weight.grad.sum(dim=1),
                    pre + "g_nkw": tr.norm_k.weight.grad})
    return out


RESIZE_CASES = [(240, 320, 256), (97, 131, 37), (64, 48, 100), (300, 256, 256)]     # (H, W, Resize(size)); last: short side already == size


def gen_resize(mods):
    """the reference's frame Resize (functional.resize_clip -> PIL.Image.resize(BILINEAR)) run with PIL itself"""
    from PIL import Image
    fn = _load("ref_functional", os.path.join(REF, "videotransforms/functional.py"))
    out = {}
    rng = np.random.default_rng(5)
    for i, (h, w, size) in enumerate(RESIZE_CASES):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[: h // 3] = (img[: h // 3] // 32) * 32            # some flat / banded areas next to the noise
        res = fn.resize_clip([Image.fromarray(img)], size, interpolation="nearest")[0]     # Resize()'s default argument
        out["c%d_in" % i] = img
        out["c%d_out" % i] = np.asarray(res)
        out["c%d_size" % i] = size
    return out


def load_reference_teacher():
    """teacher/code/model.py as a module.  Its `from utils import ...` / `from transformer import ...` mean the files next
    to it (teacher/code/utils.py, transformer.py), not the student's utils.py registered by load_reference()."""
    tdir = os.path.join(REF, "teacher", "code")
    for name in ("turtle", "timm"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.forward = None
            sys.modules[name] = m
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.models = types.ModuleType("torchvision.models")
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tv.models
    torch.Tensor.cuda = lambda self, *a, **k: self
    saved = {k: sys.modules.get(k) for k in ("utils", "transformer")}
    sys.modules["utils"] = _load("utils", os.path.join(tdir, "utils.py"))
    sys.modules["transformer"] = _load("transformer", os.path.join(tdir, "transformer.py"))
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mod = _load("ref_teacher_model", os.path.join(tdir, "model.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


MFM_CASES = [(7001, 2, 1), (7002, 3, 2)]       # (input seed, videos, shirt_num); weights: make_mfm_params(MFM_WEIGHT_SEED)
MFM_WEIGHT_SEED = 77


def gen_mfm(mods):
    """ThreeTRXShiftLoopTime.extract_feature (teacher/code/model.py:1648-1664) of the reference itself at its fixed width 2048
    (541 M parameters), eval mode + no_grad as extract_multi_feature.py:113-121 runs it; also the three partial features."""
    import warnings
    M = load_reference_teacher()
    args = argparse.Namespace(seq_len=8, trans_num=2, shirt_num=1, num_gpus=1, temp_set=[2], trans_linear_in_dim=2048,
                              trans_linear_out_dim=1152, trans_dropout=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = M.ThreeTRXShiftLoopTime(args).eval()
    p = O.make_mfm_params(MFM_WEIGHT_SEED)
    sd = model.state_dict()
    fused = [k for k in sd if k.startswith("fusion.") or k.startswith("three_fusion.")]
    assert sorted(fused) == sorted(p), "state_dict keys of the reference's fusion modules differ from oracle.mfm_param_shapes()"
    with torch.no_grad():
        for k in fused:
            sd[k].copy_(p[k])
    out = {"weight_seed": MFM_WEIGHT_SEED}
    for case, (seed, n, shirt) in enumerate(MFM_CASES):
        args.shirt_num = shirt
        rgb, depth, flow = O.make_mfm_inputs(seed, n)
        with torch.no_grad():
            y = model.extract_feature({"rgb": rgb, "depth": depth, "flow": flow})
            f1 = model.three_fusion.extract_feature(rgb, depth, flow)
            f2 = model.fusion.extract_feature(rgb, torch.roll(depth, -shirt, 1))
        pre = "c%d_" % case
        out.update({pre + "seed": seed, pre + "n": n, pre + "shirt": shirt, pre + "out": y, pre + "three": f1, pre + "two_depth": f2})
    return out


def head_case_inputs(seed, ns, nq, hw=7):
    """Re-creates the exact inputs / weights of gen_head (used by tests; no reference needed): trunk output maps [F, 512, hw, hw]
    for the support and the query frames (post-ReLU: non-negative, many exact zeros, ties for the max pooling), labels, upstream
    weights for the scalar that is differentiated, and the seeded student parameters (fc1, fc2, TRX heads)."""
    g = torch.Generator().manual_seed(7000 + seed)
    fm_s = torch.relu(torch.randn(ns * 8, 512, hw, hw, generator=g))
    fm_q = torch.relu(torch.randn(nq * 8, 512, hw, hw, generator=g))
    lab = torch.arange(5).repeat_interleave(ns // 5)[torch.randperm(ns, generator=g)].float()
    params = O.make_student_params(7100 + seed)
    params = {k: v for k, v in params.items() if not k.startswith("backbone.resnet.")}
    return fm_s, fm_q, lab, params


def gen_head(mods):
    """A3 + A1: the reference's OWN resnet18_2fc.forward (model/backbone/resnet18_2fc.py:37-77: adaptive max pool 4x4, patch mean,
    fc1 / fc2, reshape) and Student.forward (model/model_select.py:26-36: backbone -> classifier -> dict) run here.  torchvision is
    absent from this image and from the reference tree, so `torchvision.models.resnet18` is a test-only stub whose children()[:-2]
    is an identity: the 'frames' handed to the reference ARE trunk output maps [F, 512, h, w] (seeded, head_case_inputs) - every
    line after `self.resnet(...)` is the reference's code.  7x7 maps (224^2 frames) and 3x3 maps (96^2: adaptive windows that
    overlap)."""
    import torch.nn as nn
    tvm = sys.modules["torchvision.models"]

    class _StubResNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.trunk = nn.Identity()
            self.avgpool = nn.Identity()      # children()[-2:], dropped by resnet18_2fc.py:30-33
            self.fc = nn.Identity()
    tvm.resnet18 = lambda pretrained=True: _StubResNet()
    MS = _load("ref_model_select", os.path.join(REF, "model/model_select.py"))
    out = {}
    for case, (seed, ns, nq, hw) in enumerate([(1, 5, 5, 7), (2, 10, 5, 3)]):
        args = argparse.Namespace(way=5, shot=ns // 5, query_per_class=nq // 5, seq_len=8, trans_linear_out_dim=1152,
                                  trans_linear_in_dim=2048, trans_dropout=0.0, num_gpus=1, model_backbone="resnet18_2fc",
                                  model_classifier="TRX_2fcsup", temp_set=[2])
        student = MS.Student(args)
        fm_s, fm_q, lab, params = head_case_inputs(seed, ns, nq, hw)
        sd = student.state_dict()
        for k, v in params.items():
            sd[k].copy_(v)
        fm_s.requires_grad_()
        fm_q.requires_grad_()
        r = student(fm_s, lab, fm_q)
        cf, tf, lg = r["context_features"], r["target_features"], r["logits"]
        w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        wf = torch.linspace(-1, 1, 2048)
        scalar = ((lg["kl"] * w).sum() * 1e-2 + (lg["ce"] * w.flip(0)).sum() * 1e-2 + (lg["sup"] * torch.linspace(1, -1, 20).reshape(5, 4)).sum() * 1e-3
                  + (cf["context_features_1"] * wf).sum() * 1e-3 + (tf["target_features_2"] * wf.flip(0)).sum() * 1e-3)
        scalar.backward()
        pre = "c%d_" % case
        bb = student.backbone
        out.update({pre + "seed": seed, pre + "ns": ns, pre + "nq": nq, pre + "hw": hw,
                    # [N, 8, 2048] features as 64-wide chunk sums + the first video in full (keeps the fixture small)
                    pre + "cf1": gsum(cf["context_features_1"]), pre + "cf2": gsum(cf["context_features_2"]),
                    pre + "tf1": gsum(tf["target_features_1"]), pre + "tf2": gsum(tf["target_features_2"]),
                    pre + "cf1_v0": cf["context_features_1"][0].clone(), pre + "tf2_v0": tf["target_features_2"][0].clone(),
                    pre + "kl": lg["kl"], pre + "ce": lg["ce"], pre + "sup": lg["sup"],
                    # gradients: the trunk-output maps as per-frame / per-channel sums + the first frame in full, fc parameters as row sums
                    pre + "g_fm_s_fc": fm_s.grad.sum((2, 3)), pre + "g_fm_q_fc": fm_q.grad.sum((2, 3)),
                    pre + "g_fm_s_f0": fm_s.grad[0].clone(), pre + "g_fm_q_f0": fm_q.grad[0].clone(),
                    pre + "g_fc1_w": bb.fc1.weight.grad.sum(1), pre + "g_fc1_b": bb.fc1.bias.grad,
                    pre + "g_fc2_w": bb.fc2.weight.grad.sum(1), pre + "g_fc2_b": bb.fc2.bias.grad,
                    pre + "keys": np.array(sorted(r.keys()) + sorted(cf.keys()) + sorted(tf.keys()) + sorted(lg.keys()))})
    return out


def trx_case_inputs(seed, ns, nq, shuffle):
    """Re-creates the exact inputs/weights of gen_trx (used by tests; no reference needed)."""
    g = torch.Generator().manual_seed(1000 + seed)
    p = O.make_trx_params(g)
    p["norm_k.weight"] = 1 + 0.1 * torch.randn(1152, generator=g)
    p["norm_k.bias"] = 0.1 * torch.randn(1152, generator=g)
    sup1, lab, qry1 = feature_case(seed, ns, nq, 0.5, shuffle)
    sup2, _, qry2 = feature_case(seed + 100, ns, nq, 0.5, shuffle)
    return p, sup1, qry1, sup2, qry2, lab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present: fixtures can only be (re)generated in the build container")
    torch.manual_seed(0)
    mods = load_reference()
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in (("distill", gen_distill), ("distill_methods", gen_distill_methods), ("edist", gen_edist), ("trx", gen_trx),
                     ("trx_sup", gen_trx_sup), ("resize", gen_resize), ("mfm", gen_mfm), ("kl_feature", gen_kl_feature),
                     ("head", gen_head)):
        data = fn(mods)
        data = t2n(data) if name != "resize" else data
        path = os.path.join(GOLD, name + ".npz")
        if a.check_only:
            old = np.load(path)
            for k in data:
                if np.asarray(data[k]).dtype.kind in "USO":      # provenance strings
                    assert str(old[k]) == str(np.asarray(data[k])), k
                else:
                    np.testing.assert_allclose(old[k], data[k], rtol=1e-5, atol=1e-6, err_msg=k)
            print("checked", path)
        else:
            np.savez_compressed(path, **data)
            print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
