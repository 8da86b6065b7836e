"""ResNet-50 backbones (reference: model/backbone/resnet50_2fc.py:14-88, resnet50_student.py:7-60)."""
import torch.nn as nn

from ... import ops
from .resnet import Linear, ResNet50Trunk, two_trunk_calls


class resnet50_2fc(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.args.trans_linear_in_dim = 2048
        self.num_patches = 16
        self.resnet = ResNet50Trunk()
        self.fc1 = Linear(2048, 2048)
        self.fc2 = Linear(2048, 2048)

    def forward(self, context_feature, context_labels, target_feature):
        cf, tf = two_trunk_calls(self.resnet, ops.PoolHeadFn.apply, context_feature, target_feature)     # :40-57
        L, D = self.args.seq_len, self.args.trans_linear_in_dim
        return ({"context_features_1": self.fc1(cf).reshape(-1, L, D), "context_features_2": self.fc2(cf).reshape(-1, L, D)},
                {"target_features_1": self.fc1(tf).reshape(-1, L, D), "target_features_2": self.fc2(tf).reshape(-1, L, D)})

    def distribute_model(self):
        return None


class resnet50_stduent(nn.Module):
    """(sic: the reference spells it `resnet50_stduent`) pooled 2048-d trunk features, no extra head"""

    def __init__(self, args):
        super().__init__()
        self.train()
        self.args = args
        self.args.trans_linear_in_dim = 2048
        self.num_patches = 16
        self.resnet = ResNet50Trunk()

    def forward(self, context_feature, context_labels, target_feature):
        cf, tf = two_trunk_calls(self.resnet, ops.PoolHeadFn.apply, context_feature, target_feature)
        L, D = self.args.seq_len, self.args.trans_linear_in_dim
        return cf.reshape(-1, L, D), tf.reshape(-1, L, D)

    def distribute_model(self):
        return None
