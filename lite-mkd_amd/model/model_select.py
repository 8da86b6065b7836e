"""Plugin registry: Student / Teacher wrappers and name -> plugin maps
(reference: model/model_select.py:17-57,138-153,161-241).  Same string keys, same wrapper attribute
names (`backbone`, `classifier`, `backbone.resnet`, `backbone.fc1/fc2`, `classifier.transformers.*`)
so state_dict keys match the reference's checkpoints."""
import torch
import torch.nn as nn

from .. import ops
from . import classifiers
from .backbone import resnet18_2fc, resnet18_student, resnet50_2fc, resnet50_stduent

# every key of the reference's name2backbone (model_select.py:167-180); None = plugin outside the hot path
name2backbone = {
    "resnet18_student": resnet18_student,
    "resnet50_student": resnet50_stduent,
    "strm18_student": None,
    "resnet18_2fc": resnet18_2fc,
    "resnet50_2fc": resnet50_2fc,
    "strmbackbone": None,
    "meta_baseline": None,
    "meta_baseline_fc2": None,
    "moblienetv3_fc2": None,
    "moblienetv3": None,
}

# model_select.py:182-199
name2classifier = {
    "cos": "CosDistance", "TRX": "TRX", "TRX_sup": "TRX_sup", "CTX": "CTX", "TRX_2fc": "TRX_2fc",
    "TRX_1fc_sup": "TRX_1fc_sup", "TRX_2fcsup": "TRX_2fcsup", "TRX_2fcsup_2": "TRX_2fcsup_2",
    "strmclassifiers": "strmclassifiers", "e_dist": "e_dist", "e_dist_fc2": "e_dist_fc2",
    "e_dist_fc2_sup": "e_dist_fc2_sup", "strm_res18": "strmclassifiers_resnet18",
    "strm_res18_sup": "strmclassifiers_resnet18_sup", "strm_1fc_sup": "strm_1fc_sup",
    "e_dist_1fc_sup": "e_dist_1fc_sup",
}

# model_select.py:220-233
name2teacher = {
    "cos": "CosDistance", "e_dist": "e_dist", "e_dist_fc2_sup": "e_dist_fc2_sup_fixed",
    "train_teacher": "TRX", "test_teacher": "TRX_fixed",
    "train_teacher_TRX_sup": "TRX_sup", "test_teacher_TRX_sup_fixed": "TRX_sup_fixed",
    "train_teacher_TRX_2fcsup": "TRX_2fcsup", "test_teacher_TRX_2fcsup_fixed": "TRX_2fcsup_fixed",
}


def _classifier_class(classifiername):
    cls = getattr(classifiers, classifiername, None)
    if cls is None:
        raise NotImplementedError("classifier plugin '%s' is outside the MI355X hot path (SURVEY.md 8f N2)" % classifiername)
    return cls


def select_model_student(args):
    """model_select.py:161-209.  Unknown names raise KeyError like the reference."""
    backbone_cls = name2backbone[args.model_backbone]
    if backbone_cls is None:
        raise NotImplementedError("backbone plugin '%s' is outside the MI355X hot path (SURVEY.md 8f N1)" % args.model_backbone)
    backbone = backbone_cls(args)
    classifier = _classifier_class(name2classifier[args.model_classifier])(args)
    # args.num_gpus > 1: the reference wraps backbone.resnet in nn.DataParallel (:205-207); here extra GPUs
    # run whole episodes in their own processes (parallel.py), the model itself is unchanged.
    return backbone, classifier


def select_model_teacher(args):
    """model_select.py:211-241.  `test_teacher_TRX_2fcsup_fixed` loads no checkpoint in the reference (:238)."""
    classifier = _classifier_class(name2teacher[args.model_teacher])(args)
    if args.model_teacher in ["test_teacher", "test_teacher_TRX_sup_fixed"]:
        classifier = load_teacher(classifier, args)
    return classifier


def load_teacher(teacher, args):
    """model_select.py:81-136: copy `bracnch.transformers.0.*` of an MFM checkpoint into the teacher TRX."""
    sd = torch.load(args.teacher_checkpoint, map_location="cpu")["model_state_dict"]
    t = teacher.transformers
    pre = "bracnch.transformers.0."
    with torch.no_grad():
        t.pe.pe.copy_(sd[pre + "pe.pe"])
        for mod, name in ((t.k_linear, "k_linear"), (t.v_linear, "v_linear"), (t.norm_k, "norm_k"), (t.norm_v, "norm_v")):
            mod.weight.copy_(sd[pre + name + ".weight"])
            mod.bias.copy_(sd[pre + name + ".bias"])
    return teacher


def strip_dataparallel_prefix(state_dict):
    """model_select.py:143-150: `backbone.resnet.module.X` -> `backbone.resnet.X`."""
    out = {}
    for key, v in state_dict.items():
        parts = key.split(".")
        if len(parts) > 2 and parts[2] == "module":
            key = ".".join(parts[:2] + parts[3:])
        out[key] = v
    return out


def load_student(args):
    """model_select.py:138-153."""
    student = Student(args)
    ckpt = torch.load(args.test_model_path, map_location="cpu")
    student.load_state_dict(strip_dataparallel_prefix(ckpt["model_state_dict"]))
    return student


class Student(nn.Module):
    """model_select.py:17-36."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.backbone, self.classifier = select_model_student(args)

    def forward(self, context_feature, context_labels, target_feature):
        # one tiny D2H copy of the labels, before the backbone kernels are queued
        ops.get_plan(context_labels, self.args.way)
        context_features, target_features = self.backbone(context_feature, context_labels, target_feature)
        logits = self.classifier(context_features, context_labels, target_features)["logits"]
        return {"logits": logits, "context_features": context_features, "target_features": target_features}


class Teacher(nn.Module):
    """model_select.py:38-57."""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.classifier = select_model_teacher(args)

    def forward(self, context_feature, context_labels, target_feature):
        return self.classifier(context_feature, context_labels, target_feature)

    def distribute_model(self):
        return None
