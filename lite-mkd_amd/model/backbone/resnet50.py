"""ResNet-50 backbones (reference: model/backbone/resnet50_2fc.py:14-88, resnet50_student.py:7-60)."""
import torch.nn as nn

from ... import ops
from .resnet import Linear, ResNet50Trunk, trunk_features, two_trunk_calls


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
        X, Fs = trunk_features(self.resnet, ops.PoolHeadFn.apply, context_feature, target_feature)     # :40-57
        L, D = self.args.seq_len, self.args.trans_linear_in_dim
        c1, t1, c2, t2 = ops.two_head_linear_x(X, Fs, self.fc1, self.fc2)
        return ({"context_features_1": c1.reshape(-1, L, D), "context_features_2": c2.reshape(-1, L, D)},
                {"target_features_1": t1.reshape(-1, L, D), "target_features_2": t2.reshape(-1, L, D)})

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
