"""resnet18_student backbone (reference: model/backbone/resnet18_student.py:15-69): single 512->2048 head."""
import torch.nn as nn

from ... import ops
from .resnet import Linear, ResNet18Trunk, two_trunk_calls


class resnet18_student(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.args.trans_linear_in_dim = 2048
        self.num_patches = 16
        self.resnet = ResNet18Trunk()
        self.res18_2048 = Linear(512, 2048)

    def forward(self, context_feature, context_labels, target_feature):
        cf, tf = two_trunk_calls(self.resnet, ops.PoolHeadFn.apply, context_feature, target_feature)
        L, D = self.args.seq_len, self.args.trans_linear_in_dim
        return self.res18_2048(cf).reshape(-1, L, D), self.res18_2048(tf).reshape(-1, L, D)

    def distribute_model(self):
        return None
