"""Euclidean matcher family (reference: model/classifiers/e_dist_fc2.py:46-231, e_dist.py:23-62)."""
import torch
import torch.nn as nn

from ... import ops
from .TRX_2fcsup import SupportDK


class e_dist(nn.Module):
    """e_dist_fc2.py:46-91: frame-mean embeddings, cdist(p=2) to every shot, mean over the class, negated."""

    def __init__(self, args):
        super().__init__()
        self.args = args

    def forward(self, support_set, support_labels, queries):
        supports = support_set.reshape(-1, 8, 2048)
        queries = queries.reshape(-1, 8, 2048)
        plan = ops.get_plan(support_labels, self.args.way)
        return {"logits": ops.EDistFn.apply(supports, queries, plan)}


class e_dist_fc2(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.e_dict = e_dist(args)

    def forward(self, context_feature, context_labels, target_feature):
        l1 = self.e_dict(context_feature["context_features_1"], context_labels, target_feature["target_features_1"])["logits"]
        l2 = self.e_dict(context_feature["context_features_2"], context_labels, target_feature["target_features_2"])["logits"]
        return {"logits": {"fc_1": l1, "fc_2": l2}}


class e_dist_fc2_sup(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.e_dict = e_dist(args)
        self.supportKD = SupportDK(args)

    def forward(self, context_feature, context_labels, target_feature):
        l1 = self.e_dict(context_feature["context_features_1"], context_labels, target_feature["target_features_1"])["logits"]
        l2 = self.e_dict(context_feature["context_features_2"], context_labels, target_feature["target_features_2"])["logits"]
        l3 = self.supportKD(context_feature["context_features_2"], context_labels, target_feature["target_features_2"])["logits"]
        return {"logits": {"kl": l1, "ce": l2, "sup": l3}}


class e_dist_1fc_sup(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.e_dict = e_dist(args)
        self.supportKD = SupportDK(args)

    def forward(self, context_feature, context_labels, target_feature):
        l2 = self.e_dict(context_feature, context_labels, target_feature)["logits"]
        l3 = self.supportKD(context_feature, context_labels, target_feature)["logits"]
        return {"logits": {"kl": l2, "sup": l3}}


class e_dist_fc2_sup_fixed(nn.Module):
    """e_dist_fc2.py:203-231 — teacher-side variant (the reference does not wrap it in no_grad; teacher
    features carry no grad anyway)."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.e_dict = e_dist(args)
        self.supportKD = SupportDK(args)

    def forward(self, context_feature, context_labels, target_feature):
        l1 = self.e_dict(context_feature, context_labels, target_feature)["logits"]
        l2 = self.supportKD(context_feature, context_labels, target_feature)["logits"]
        return {"logits": {"kl": l1, "sup": l2}}


class CosDistance(e_dist):
    """model/classifiers/COS.py:23-62: despite the name the same Euclidean matcher, but it returns the bare [Nq, way] tensor
    (so, as in the reference, it cannot be used through Student, which indexes ['logits'])."""

    def forward(self, support_set, support_labels, queries):
        return super().forward(support_set, support_labels, queries)["logits"]
