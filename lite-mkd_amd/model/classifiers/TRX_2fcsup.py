"""TRX_2fcsup / TRX_2fcsup_fixed classifiers (reference: model/classifiers/TRX_2fcsup.py:24-256).

Differences from the reference that do not change results: logits stay on the device (the
reference assembles them in CPU tensors, :114,180, and the distiller moves them back); the
4096-wide tuple Linear is evaluated as two 2048-wide per-frame projections (same sum, 3.5x fewer
FLOPs); all classes are matched in one pass over class-sorted support tuples."""
import math

import torch
import torch.nn as nn

from ... import ops
from ...parallel import rank as _rank


class PositionalEncoding(nn.Module):
    """TRX_2fcsup.py:24-48.  Holds the `pe` buffer [1,max_len,d_model]; the add (+dropout) is fused into
    the TRX op."""

    def __init__(self, d_model, dropout, max_len=5000, pe_scale_factor=0.1):
        super().__init__()
        self.p = float(dropout)
        self.pe_scale_factor = pe_scale_factor
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term) * self.pe_scale_factor
        pe[:, 1::2] = torch.cos(position * div_term) * self.pe_scale_factor
        self.register_buffer("pe", pe.unsqueeze(0))


class TemporalCrossTransformer(nn.Module):
    """TRX_2fcsup.py:50-160.  Parameters are held by standard nn.Linear / nn.LayerNorm containers so the
    state_dict keys (k_linear.*, v_linear.*, norm_k.*, norm_v.*, pe.pe) and default init match."""

    def __init__(self, args, temporal_set_size=2):
        super().__init__()
        if temporal_set_size != 2:
            raise NotImplementedError("only frame pairs (temporal_set_size=2) are on the hot path")
        self.args = args
        self.temporal_set_size = temporal_set_size
        max_len = int(self.args.seq_len * 1.5)
        self.pe = PositionalEncoding(2048, self.args.trans_dropout, max_len=max_len)
        self.k_linear = nn.Linear(2048 * temporal_set_size, self.args.trans_linear_out_dim)
        self.v_linear = nn.Linear(2048 * temporal_set_size, self.args.trans_linear_out_dim)
        self.norm_k = nn.LayerNorm(self.args.trans_linear_out_dim)
        self.norm_v = nn.LayerNorm(self.args.trans_linear_out_dim)       # unused by forward (:106), as in the reference
        self.tuples_len = self.args.seq_len * (self.args.seq_len - 1) // 2

    @staticmethod
    def draw_dropout_seed():
        """One 62-bit draw from torch's default (CPU) generator per dropout call, mixed with the distributed rank: every call of
        every module (student heads, frozen teacher) gets an independent mask, as with the reference's nn.Dropout on the global
        Philox stream (TRX_2fcsup.py:28,48); ranks that share one torch.manual_seed for equal initial weights still draw
        different masks for their different episodes; torch.manual_seed / get_rng_state / set_rng_state reproduce and
        checkpoint the sequence.  No device work, no synchronisation."""
        s = int(torch.randint(0, 1 << 62, (1,)).item())
        return (s ^ (_rank() * 0x9E3779B97F4A7C15)) & 0x3FFFFFFFFFFFFFFF

    def _mask(self, n_rows, device):
        p = self.pe.p
        if not self.training or p <= 0.0:
            return None
        # while a hipGraph is being captured the kernel reads its seed from a device slot (ops.SEED_SLOTS) that the replay loop fills:
        # the capture itself must not consume a draw, or the replays' seed sequence would lag the eager loop's by one episode
        seed = 0 if ops.SEED_SLOTS is not None else self.draw_dropout_seed()
        return ops.dropout_mask((n_rows, 2048), p, seed, device)

    def forward(self, support_set, support_labels, queries, with_sim=False):
        L = self.args.seq_len
        if support_set.shape[1] != L or support_set.shape[2] != 2048:
            raise RuntimeError("TemporalCrossTransformer expects [N,%d,2048] features" % L)
        plan = ops.get_plan(support_labels, self.args.way)
        pe = self.pe.pe[0, :L].contiguous()
        mask = self._mask((support_set.shape[0] + queries.shape[0]) * L, support_set.device)
        a = (self.k_linear.weight, self.k_linear.bias, self.v_linear.weight, self.v_linear.bias,
             self.norm_k.weight, self.norm_k.bias, pe, mask, plan)
        grad = torch.is_grad_enabled() and (support_set.requires_grad or self.k_linear.weight.requires_grad)
        if with_sim:      # TRX_sup.py:74-178: + cosine similarity between each query's class prototypes
            logits, sim = ops.TRXSupFn.apply(support_set, queries, *a) if grad else ops.trx_sup_nograd(support_set, queries, plan, *a[:8])
            return {"logits": {"support_set": sim, "query": logits}}
        if grad:
            logits = ops.TRXLogitsFn.apply(support_set, queries, *a)
        else:
            logits = ops.trx_logits_nograd(support_set, queries, plan, *a[:8])
        return {"logits": logits}


class SupportDK(nn.Module):
    """TRX_2fcsup.py:162-189 (== e_dist_fc2.py:17-44).  Ignores the labels and assumes class-sorted
    support, exactly like the reference; output is [5,4] (way x way-1)."""

    def __init__(self, args):
        super().__init__()
        self.args = args

    def forward(self, support_set, support_labels, queries):
        return {"logits": ops.SupportDKFn.apply(support_set, self.args.way, self.args.shot)}


class TRX_2fcsup(nn.Module):
    """TRX_2fcsup.py:191-224."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)
        self.supportKD = SupportDK(args)

    def forward(self, context_feature, context_labels, target_feature):
        c1, t1 = context_feature["context_features_1"], target_feature["target_features_1"]
        c2, t2 = context_feature["context_features_2"], target_feature["target_features_2"]
        l1 = self.transformers(c1, context_labels, t1)["logits"]
        if ops.HEADS_ON_TWO_STREAMS and c2.is_cuda:
            # the 'ce' head does not depend on the 'kl' head: its ~20 small GEMM / softmax launches (126 workgroups each on a 256-CU
            # chip) run on the auxiliary stream beside the first head's, forward and - autograd replays a node on the stream of its
            # forward - backward
            main, aux = torch.cuda.current_stream(c2.device), ops.aux_stream(c2.device)
            ops.get_plan(context_labels, self.args.way).full_rowmap(c2.shape[0] + t2.shape[0])      # shared per-episode tensors: made on main
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                l2 = self.transformers(c2, context_labels, t2)["logits"]
            for t in (c2, t2):
                t.record_stream(aux)
            l3 = self.supportKD(c2, context_labels, t2)["logits"]
            main.wait_stream(aux)
            l2.record_stream(main)
        else:
            l2 = self.transformers(c2, context_labels, t2)["logits"]
            l3 = self.supportKD(c2, context_labels, t2)["logits"]
        return {"logits": {"kl": l1, "ce": l2, "sup": l3}}


class TRX_2fcsup_fixed(nn.Module):
    """TRX_2fcsup.py:226-256 — the frozen teacher head (no_grad)."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)
        self.supportKD = SupportDK(args)

    def forward(self, context_feature, context_labels, target_feature):
        with torch.no_grad():
            l1 = self.transformers(context_feature, context_labels, target_feature)["logits"]
            l2 = self.supportKD(context_feature, context_labels, target_feature)["logits"]
        return {"logits": {"kl": l1, "sup": l2}}


class TRX(nn.Module):
    """model/classifiers/TRX.py:167-183 — one TemporalCrossTransformer head, returns {'logits': [Nq, way]}."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)

    def forward(self, context_feature, context_labels, target_feature):
        return self.transformers(context_feature, context_labels, target_feature)


class TRX_fixed(nn.Module):
    """model/classifiers/TRX.py:186-211 — frozen single-head teacher (weights come from load_teacher)."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)

    def forward(self, context_feature, context_labels, target_feature):
        with torch.no_grad():
            L, D = self.args.seq_len, self.args.trans_linear_in_dim
            c = context_feature.reshape(-1, L, D)
            t = target_feature.reshape(-1, L, D)
            return {"logits": self.transformers(c, context_labels, t)["logits"]}


class TRX_sup(nn.Module):
    """model/classifiers/TRX_sup.py:194-209 — {'logits': {'support_set': [Nq, way, way] prototype cosine similarities,
    'query': [Nq, way] TRX logits}} (consumed by Distiller.support_sim)."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)

    def forward(self, context_feature, context_labels, target_feature):
        return self.transformers(context_feature, context_labels, target_feature, with_sim=True)


class TRX_sup_fixed(nn.Module):
    """model/classifiers/TRX_sup.py:212-229 — the same under no_grad (frozen teacher)."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)

    def forward(self, context_feature, context_labels, target_feature):
        with torch.no_grad():
            return self.transformers(context_feature, context_labels, target_feature, with_sim=True)


class TRX_2fc(nn.Module):
    """model/classifiers/TRX_2fc.py:163-192 — the two fc heads through the shared transformer, {'fc_1','fc_2'}."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.transformers = TemporalCrossTransformer(args, 2)

    def forward(self, context_feature, context_labels, target_feature):
        l1 = self.transformers(context_feature["context_features_1"], context_labels, target_feature["target_features_1"])["logits"]
        l2 = self.transformers(context_feature["context_features_2"], context_labels, target_feature["target_features_2"])["logits"]
        return {"logits": {"fc_1": l1, "fc_2": l2}}
