"""CPU: the oracle restatement (oracle/ref_cpu.py) against the golden vectors produced by
the reference's own modules (oracle/gen_golden.py, run in the build container)."""
import os

import numpy as np
import torch

from oracle import ref_cpu as O
from oracle.gen_golden import feature_case, gsum, head_case_inputs, trx_case_inputs

RTOL, ATOL = 2e-5, 2e-5     # fp32, same library (torch CPU) on both sides


def T(x):
    return torch.from_numpy(np.asarray(x))


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, np.asarray(b), rtol=rtol, atol=atol)


def test_distill_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "distill.npz"))
    for c in range(3):
        pre = "c%d_" % c
        s = {k: T(G[pre + "s_" + k]).clone().requires_grad_() for k in ("kl", "ce", "sup")}
        t = {k: T(G[pre + "t_" + k]) for k in ("kl", "sup")}
        labels = T(G[pre + "labels"])
        r = O.distill_fc_2_sup_dist(s, t, labels)
        r["loss"].backward()
        close(r["loss"], G[pre + "loss"])
        close(r["soft_loss"], G[pre + "soft"])
        close(r["hard_loss"], G[pre + "hard"])
        close(O.kd_loss(s["kl"], t["kl"], 4), G[pre + "kd"])
        close(O.inter_class_relation(s["sup"], t["sup"]), G[pre + "icr"])
        for k in ("kl", "ce", "sup"):
            close(s[k].grad, G[pre + "g_" + k], atol=1e-7)
        s1 = T(G[pre + "s_kl"]).clone().requires_grad_()
        r = O.distill_KD(s1, t["kl"], labels)
        r["loss"].backward()
        close(r["loss"], G[pre + "KD_loss"])
        close(s1.grad, G[pre + "KD_g"], atol=1e-7)
        s2 = T(G[pre + "s_kl"]).clone().requires_grad_()
        r = O.distill_Dist_KD(s2, t["kl"], labels)
        r["loss"].backward()
        close(r["loss"], G[pre + "DistKD_loss"])
        close(s2.grad, G[pre + "DistKD_g"], atol=1e-7)
        lg = s["kl"].detach() + s["ce"].detach()
        assert np.array_equal(torch.argmax(lg, -1).numpy(), G[pre + "argmax"])      # bit-exact indices
        close(O.aggregate_accuracy(lg, labels), G[pre + "acc"], 0, 0)


def test_edist_supportdk_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "edist.npz"))
    for c in range(3):
        pre = "c%d_" % c
        ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
        sup, lab, qry = feature_case(int(G[pre + "seed"]), ns, nq, 1.0, bool(G[pre + "shuffle"]))
        sup.requires_grad_()
        qry.requires_grad_()
        r = O.clf_e_dist_1fc_sup(sup, lab, qry, 5, ns // 5)
        close(r["kl"], G[pre + "kl"])
        close(r["sup"], G[pre + "sup"], rtol=1e-5, atol=1e-3)
        w_kl = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        w_sup = torch.linspace(1, -1, 20).reshape(5, 4)
        ((r["kl"] * w_kl).sum() + (r["sup"] * w_sup).sum() * 1e-3).backward()
        close(gsum(sup.grad), G[pre + "g_sup_feat"], atol=1e-5)
        close(gsum(qry.grad), G[pre + "g_qry_feat"], atol=1e-5)
        close(sup.grad[0, :, :128], G[pre + "g_sup_row0"], atol=1e-6)


def test_trx_golden(golden_dir):
    G = np.load(os.path.join(golden_dir, "trx.npz"))
    for c in range(3):
        pre = "c%d_" % c
        ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
        p, sup1, qry1, sup2, qry2, lab = trx_case_inputs(int(G[pre + "seed"]), ns, nq, bool(G[pre + "shuffle"]))
        for k in ("k_linear.weight", "k_linear.bias", "v_linear.weight", "v_linear.bias", "norm_k.weight", "norm_k.bias"):
            p[k].requires_grad_()
        for x in (sup1, qry1, sup2, qry2):
            x.requires_grad_()
        r = O.clf_TRX_2fcsup({"context_features_1": sup1, "context_features_2": sup2}, lab,
                             {"target_features_1": qry1, "target_features_2": qry2}, p, 5, ns // 5)
        # logits are O(1e2..1e3) sums of squares
        close(r["kl"], G[pre + "kl"], rtol=1e-5, atol=1e-3)
        close(r["ce"], G[pre + "ce"], rtol=1e-5, atol=1e-3)
        close(r["sup"], G[pre + "sup"], rtol=1e-5, atol=1e-3)
        w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        w_sup = torch.linspace(1, -1, 20).reshape(5, 4)
        ((r["kl"] * w).sum() * 1e-2 + (r["ce"] * w.flip(0)).sum() * 1e-2 + (r["sup"] * w_sup).sum() * 1e-3).backward()
        close(gsum(sup1.grad), G[pre + "g_sup1"], rtol=1e-4, atol=1e-5)
        close(gsum(qry1.grad), G[pre + "g_qry1"], rtol=1e-4, atol=1e-5)
        close(gsum(sup2.grad), G[pre + "g_sup2"], rtol=1e-4, atol=1e-5)
        close(gsum(qry2.grad), G[pre + "g_qry2"], rtol=1e-4, atol=1e-5)
        close(p["k_linear.bias"].grad, G[pre + "g_kb"], rtol=1e-4, atol=1e-5)
        close(p["v_linear.bias"].grad, G[pre + "g_vb"], rtol=1e-4, atol=1e-5)
        close(p["k_linear.weight"].grad.sum(1), G[pre + "g_kw_sum"], rtol=1e-3, atol=1e-4)
        close(p["v_linear.weight"].grad.sum(1), G[pre + "g_vw_sum"], rtol=1e-3, atol=1e-4)
        close(p["norm_k.weight"].grad, G[pre + "g_nkw"], rtol=1e-4, atol=1e-5)
        close(p["norm_k.bias"].grad, G[pre + "g_nkb"], rtol=1e-4, atol=1e-5)
        rf = O.clf_TRX_2fcsup_fixed(sup1.detach(), lab, qry1.detach(), p, 5, ns // 5)
        close(rf["kl"], G[pre + "fixed_kl"], rtol=1e-5, atol=1e-3)
        close(rf["sup"], G[pre + "fixed_sup"], rtol=1e-5, atol=1e-3)


def test_trx_sup_golden(golden_dir):
    """TRX_sup / TRX_sup_fixed + Distiller.support_sim (reference modules run here by oracle/gen_golden.py)"""
    G = np.load(os.path.join(golden_dir, "trx_sup.npz"))
    for c in range(2):
        pre = "c%d_" % c
        ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
        p, sup, qry, sup_t, qry_t, lab = trx_case_inputs(int(G[pre + "seed"]), ns, nq, bool(G[pre + "shuffle"]))
        for k in ("k_linear.weight", "v_linear.weight", "norm_k.weight"):
            p[k].requires_grad_()
        sup.requires_grad_()
        qry.requires_grad_()
        q, sim = O.trx_sup_logits(sup, lab, qry, p)
        close(q, G[pre + "query"], rtol=1e-5, atol=1e-3)
        close(sim, G[pre + "support_set"], rtol=1e-5, atol=1e-6)
        pt = {k: (0.9 * v).detach() for k, v in p.items()}
        qt, simt = O.trx_sup_logits(sup_t, lab, qry_t, pt)
        close(qt, G[pre + "fixed_query"], rtol=1e-5, atol=1e-3)
        close(simt, G[pre + "fixed_support_set"], rtol=1e-5, atol=1e-6)
        labels = torch.arange(nq) % 5
        loss = O.distill_method("support_sim", {"query": q, "support_set": sim}, {"query": qt, "support_set": simt}, labels)
        close(loss, G[pre + "loss"], rtol=1e-5, atol=1e-6)
        w = torch.linspace(-1, 1, nq * 25).reshape(nq, 5, 5)
        (loss + (sim * w).sum() * 1e-2).backward()
        close(gsum(sup.grad), G[pre + "g_sup"], rtol=1e-4, atol=1e-5)
        close(gsum(qry.grad), G[pre + "g_qry"], rtol=1e-4, atol=1e-5)
        # row sums over 4096 weight-gradient entries of mixed sign: fp32 summation-order noise ~2e-4
        close(p["k_linear.weight"].grad.sum(1), G[pre + "g_kw_sum"], rtol=1e-3, atol=1e-3)
        close(p["v_linear.weight"].grad.sum(1), G[pre + "g_vw_sum"], rtol=1e-3, atol=1e-3)
        close(p["norm_k.weight"].grad, G[pre + "g_nkw"], rtol=1e-4, atol=1e-5)


def test_resize_golden(golden_dir):
    """frame Resize: the numpy restatement of Pillow's 8-bit BILINEAR resampler vs fixtures made by the reference's
    functional.resize_clip with PIL itself — bit exact, incl. the 'short side already matches' early return."""
    G = np.load(os.path.join(golden_dir, "resize.npz"))
    for c in range(4):
        img, ref, size = G["c%d_in" % c], G["c%d_out" % c], int(G["c%d_size" % c])
        oh, ow = O.resize_short_side(img.shape[0], img.shape[1], size)
        assert (oh, ow) == ref.shape[:2]
        assert np.array_equal(O.pil_resize_bilinear_u8(img, ow, oh), ref)
    assert O.resize_short_side(240, 320, 256) == (256, 341) and O.resize_short_side(300, 256, 256) == (300, 256)


def test_trunk_shapes_and_param_count():
    shp = O.resnet18_trunk_param_shapes()
    n = sum(int(np.prod(s)) for k, s in shp.items() if "running" not in k and "num_batches" not in k)
    assert n == 11176512
    g = torch.Generator().manual_seed(0)
    sd = O.init_resnet18_trunk(g)
    y = O.resnet18_trunk(torch.rand(2, 3, 64, 64, generator=g), sd)
    assert y.shape == (2, 512, 2, 2)
    assert int(sd["1.running_mean"].abs().sum() > 0)         # train-mode BN updated running stats


def test_mfm_param_count_and_shape():
    shp = O.mfm_param_shapes()
    n = sum(int(np.prod(s)) for s in shp.values())
    assert 5.0e8 < n < 5.8e8                                 # ~541 M


def test_mfm_golden(golden_dir):
    """MFM fusion restatement vs the reference's own ThreeTRXShiftLoopTime.extract_feature at its fixed width 2048
    (teacher/code/model.py:1648-1664; fixtures by oracle/gen_golden.py: gen_mfm): fused feature and the partial features of
    the three- and two-modality encoders, two cases (shirt_num 1 and 2)."""
    G = np.load(os.path.join(golden_dir, "mfm.npz"))
    p = O.make_mfm_params(int(G["weight_seed"]))
    for case in (0, 1):
        pre = "c%d_" % case
        n, shirt = int(G[pre + "n"]), int(G[pre + "shirt"])
        rgb, depth, flow = O.make_mfm_inputs(int(G[pre + "seed"]), n)
        with torch.no_grad():
            close(O.mfm_three_fusion(rgb, depth, flow, p), G[pre + "three"], 1e-4, 1e-4)
            close(O.mfm_two_fusion(rgb, torch.roll(depth, -shirt, 1), p), G[pre + "two_depth"], 1e-4, 1e-4)
            close(O.mfm_extract_feature(rgb, depth, flow, p, shirt, 2), G[pre + "out"], 1e-4, 1e-4)


def test_kl_feature_golden(golden_dir):
    """Distiller.KL_feature (distillers.py:126-150) restatement vs the reference's own method"""
    from oracle.gen_golden import kl_feature_inputs
    G = np.load(os.path.join(golden_dir, "kl_feature.npz"))
    for case in (0, 1):
        pre = "c%d_" % case
        s, t, labels = kl_feature_inputs(int(G[pre + "seed"]), int(G[pre + "nq"]), int(G[pre + "nv"]))
        s["logits"].requires_grad_()
        s["feature"].requires_grad_()
        r = O.distill_KL_feature(s, t, labels)
        r["loss"].backward()
        close(r["loss"], G[pre + "loss"], 1e-5, 1e-5)
        close(r["feature_loss"], G[pre + "feature_loss"], 1e-5, 1e-6)
        close(s["logits"].grad, G[pre + "g_logits"], 1e-4, 1e-7)
        close(gsum(s["feature"].grad), G[pre + "g_feature"], 1e-4, 1e-8)


def test_distill_methods_golden(golden_dir):
    """all 22 logits-only Distiller methods of the reference (distillers.py:42-733): value + student-logit gradients"""
    G = np.load(os.path.join(golden_dir, "distill_methods.npz"))
    for i, name in enumerate(sorted(O.DISTILL_SIGNATURES)):
        s, t, labels = O.distill_inputs(name, 500 + i)
        leaves = {"": s} if torch.is_tensor(s) else s
        for v in leaves.values():
            v.requires_grad_()
        loss = O.distill_method(name, s, t, labels)
        loss.backward()
        close(loss, G[name + "__loss"], 1e-5, 1e-5)
        for k, v in leaves.items():
            key = name + ("__g" if k == "" else "__g_" + k)
            close(v.grad if v.grad is not None else torch.zeros_like(v), G[key], 1e-4, 1e-7)


def test_mfm_oracle_matches_torch_transformer_encoder():
    """the MFM restatement vs torch's own nn.TransformerEncoder / nn.Embedding / nn.LayerNorm modules (the building
    blocks the reference instantiates at teacher/code/model.py:1300-1331,1361-1392), reduced width"""
    import torch.nn as nn
    d, L, N = 64, 8, 3
    torch.manual_seed(0)
    pes = [(nn.Embedding(L, d), nn.LayerNorm(d)) for _ in range(3)]
    enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=3 * d, nhead=3, batch_first=True), num_layers=2).eval()
    f1 = nn.Linear(3 * d, d)
    p = {}
    for i, (e, ln) in enumerate(pes):
        p["three_fusion.positionEncoding%d.position_embeddings.weight" % (i + 1)] = e.weight.detach()
        p["three_fusion.positionEncoding%d.LayerNorm.weight" % (i + 1)] = ln.weight.detach()
        p["three_fusion.positionEncoding%d.LayerNorm.bias" % (i + 1)] = ln.bias.detach()
    for k, v in enc.state_dict().items():
        p["three_fusion.transformer_encoder." + k] = v
    p["three_fusion.f1.weight"], p["three_fusion.f1.bias"] = f1.weight.detach(), f1.bias.detach()
    xs = [torch.randn(N, L, d) for _ in range(3)]
    with torch.no_grad():
        h = torch.cat([ln(x + e.weight[:L]) for x, (e, ln) in zip(xs, pes)], -1)
        ref = f1(enc(h))
        out = O.mfm_three_fusion(xs[0], xs[1], xs[2], p)
    close(out, ref, 1e-4, 1e-5)


def _close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, np.asarray(b), rtol=rtol, atol=atol, err_msg=msg)


def _head_case(G, case):
    pre = "c%d_" % case
    return pre, head_case_inputs(int(G[pre + "seed"]), int(G[pre + "ns"]), int(G[pre + "nq"]), int(G[pre + "hw"]))


def test_head_golden(golden_dir):
    """A3 + A1 pinned by the reference itself: tests/golden/head.npz holds what the reference's resnet18_2fc.forward
    (resnet18_2fc.py:44-77) and Student.forward (model_select.py:26-36) return for seeded trunk OUTPUT maps (torchvision's trunk
    stubbed by an identity, oracle/gen_golden.py gen_head).  The oracle's head_2fc + clf_TRX_2fcsup + the dict plumbing of
    student_forward must reproduce features, logits and the gradients w.r.t. the maps and the fc parameters."""
    G = np.load(os.path.join(golden_dir, "head.npz"))
    for case in (0, 1):
        pre, (fm_s, fm_q, lab, params) = _head_case(G, case)
        ns, nq = int(G[pre + "ns"]), int(G[pre + "nq"])
        fm_s.requires_grad_()
        fm_q.requires_grad_()
        p = {k: v.clone().requires_grad_() if v.is_floating_point() and not k.endswith("pe.pe") else v for k, v in params.items()}
        bp = {k[len("backbone."):]: v for k, v in p.items() if k.startswith("backbone.")}
        cp = {k[len("classifier.transformers."):]: v for k, v in p.items() if k.startswith("classifier.transformers.")}
        cf, tf = O.head_2fc(fm_s, fm_q, bp)
        lg = O.clf_TRX_2fcsup(cf, lab, tf, cp, 5, ns // 5)
        out = {"logits": lg, "context_features": cf, "target_features": tf}
        assert sorted(out.keys()) + sorted(cf.keys()) + sorted(tf.keys()) + sorted(lg.keys()) == list(G[pre + "keys"])
        for k, t in (("cf1", cf["context_features_1"]), ("cf2", cf["context_features_2"]), ("tf1", tf["target_features_1"]),
                     ("tf2", tf["target_features_2"])):
            _close(gsum(t), G[pre + k], 2e-5, 1e-5, k)
        _close(cf["context_features_1"][0], G[pre + "cf1_v0"], 2e-5, 1e-6, "cf1_v0")
        _close(tf["target_features_2"][0], G[pre + "tf2_v0"], 2e-5, 1e-6, "tf2_v0")
        for k in ("kl", "ce", "sup"):
            _close(lg[k], G[pre + k], 2e-5, 1e-3, k)
        w = torch.linspace(-1, 1, nq * 5).reshape(nq, 5)
        wf = torch.linspace(-1, 1, 2048)
        ((lg["kl"] * w).sum() * 1e-2 + (lg["ce"] * w.flip(0)).sum() * 1e-2 + (lg["sup"] * torch.linspace(1, -1, 20).reshape(5, 4)).sum() * 1e-3
         + (cf["context_features_1"] * wf).sum() * 1e-3 + (tf["target_features_2"] * wf.flip(0)).sum() * 1e-3).backward()
        _close(fm_s.grad.sum((2, 3)), G[pre + "g_fm_s_fc"], 1e-4, 1e-6, "g_fm_s")
        _close(fm_q.grad.sum((2, 3)), G[pre + "g_fm_q_fc"], 1e-4, 1e-6, "g_fm_q")
        _close(fm_s.grad[0], G[pre + "g_fm_s_f0"], 1e-4, 1e-7, "g_fm_s_f0")
        _close(fm_q.grad[0], G[pre + "g_fm_q_f0"], 1e-4, 1e-7, "g_fm_q_f0")
        for h in (1, 2):
            _close(p["backbone.fc%d.weight" % h].grad.sum(1), G[pre + "g_fc%d_w" % h], 1e-4, 1e-5, "fc%d.weight" % h)
            _close(p["backbone.fc%d.bias" % h].grad, G[pre + "g_fc%d_b" % h], 1e-4, 1e-6, "fc%d.bias" % h)
