"""resnet18_2fc backbone (reference: model/backbone/resnet18_2fc.py:16-86)."""
import torch.nn as nn

from ... import ops
from .resnet import Linear, ResNet18Trunk, trunk_features


class resnet18_2fc(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.args.trans_linear_in_dim = 2048          # resnet18_2fc.py:27
        self.num_patches = 16
        self.resnet = ResNet18Trunk()
        self.fc1 = Linear(512, 2048)
        self.fc2 = Linear(512, 2048)

    def forward(self, context_feature, context_labels, target_feature):
        # two separate trunk calls = two separate BatchNorm batches (resnet18_2fc.py:41-42)
        X, Fs = trunk_features(self.resnet, ops.PoolHeadFn.apply, context_feature, target_feature)     # :41-54: both calls, one launch per layer
        L, D = self.args.seq_len, self.args.trans_linear_in_dim
        c1, t1, c2, t2 = ops.two_head_linear_x(X, Fs, self.fc1, self.fc2)      # :56-64, one autograd node for the four calls
        context_feature_dict = {
            "context_features_1": c1.reshape(-1, L, D),
            "context_features_2": c2.reshape(-1, L, D),
        }
        target_features_dict = {
            "target_features_1": t1.reshape(-1, L, D),
            "target_features_2": t2.reshape(-1, L, D),
        }
        return context_feature_dict, target_features_dict

    def distribute_model(self):
        """The reference wraps self.resnet in nn.DataParallel (resnet18_2fc.py:80-86).  Here multi-GPU is
        episode-parallel (one process per GPU, see parallel.py), so this is a no-op."""
        return None
