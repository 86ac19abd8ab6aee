"""Model definitions mirroring the reference's model plugins line for line in LAYER CALLS
(SURVEY 8f, row N1).  Only the layers of the hot path run in HIP kernels (kgcn_amd.layers);
activations, the loss and the BatchNorm affine are elementwise torch ops around them.

  GCN  -- example_model/model.py:41-61      GraphConv(50) x3, BN, GraphDense(50), GraphGather, Dense(2)
  GIN  -- example_model/model_gin.py:40-67  2 x [GINAggregate, GraphDense(50) x2], Gather x2, Dense(2)

Keras learning-phase semantics (quirk Q6): the reference calls BatchNormalization / Dropout
without `training=`; under TF1 graph mode that is inference behaviour -- BN normalises with its
moving statistics (0, 1) and Dropout is the identity.  That is what is implemented here
(GraphBatchNormalization: y = gamma * x / sqrt(1 + 1e-3) + beta on the valid node rows, zero on the
padding rows, kgcn/layers.py:196-215).
"""
import math

import torch
from torch import nn

from . import layers, ops


class GraphBatchNormalization(nn.Module):
    """kgcn/layers.py:170-220 in inference mode (moving mean 0, variance 1, epsilon 1e-3)."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps
        self.gamma = None
        self.beta = None

    def forward(self, x, max_node_num=None, enabled_node_nums=None):
        if self.gamma is None:
            self.gamma = nn.Parameter(torch.ones(x.shape[-1], device=x.device))
            self.beta = nn.Parameter(torch.zeros(x.shape[-1], device=x.device))
        y = x * (self.gamma / math.sqrt(1.0 + self.eps)) + self.beta
        if enabled_node_nums is not None:       # valid rows only; padding rows are zero
            n = x.shape[1]
            en = torch.as_tensor(enabled_node_nums, device=x.device).reshape(-1, 1)
            y = y * (torch.arange(n, device=x.device).reshape(1, n) < en).to(y.dtype).unsqueeze(-1)
        return y


class KerasDense(nn.Module):
    """K.layers.Dense(units) on [B, D] (model.py:55): kernel glorot-uniform, bias zeros."""

    def __init__(self, units):
        super().__init__()
        self.units = units
        self.kernel = None
        self.bias = None

    def forward(self, x):
        if self.kernel is None:
            self.kernel = nn.Parameter(layers._init_tensor((x.shape[1], self.units), "glorot_uniform", x.device))
            self.bias = nn.Parameter(torch.zeros(self.units, device=x.device))
        return ops.dense(x, self.kernel, self.bias)


def masked_softmax_ce(logits, labels, mask):
    """model.py:56-61: cost = mask * softmax_cross_entropy(labels, logits);
    cost_opt = reduce_mean(cost) over the PADDED batch (quirk Q5); cost_sum = reduce_sum(cost)."""
    logp = torch.log_softmax(logits, dim=1)
    cost = mask * -(labels.to(logits.dtype) * logp).sum(dim=1)
    return cost.mean(), cost.sum()


class GCN(nn.Module):
    """example_model/model.py:30-71."""

    def __init__(self, adj_channel_num=1, num_classes=2):
        super().__init__()
        self.conv1 = layers.GraphConv(50, adj_channel_num)
        self.conv2 = layers.GraphConv(50, adj_channel_num)
        self.conv3 = layers.GraphConv(50, adj_channel_num)
        self.bn = GraphBatchNormalization()
        self.dense = layers.GraphDense(50)
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None):
        layer = self.conv1(features, adj=adjs)
        layer = torch.sigmoid(layer)
        layer = self.conv2(layer, adj=adjs)
        layer = torch.sigmoid(layer)
        layer = self.conv3(layer, adj=adjs)
        layer = self.bn(layer, max_node_num=features.shape[1], enabled_node_nums=enabled_node_nums)
        layer = torch.sigmoid(layer)
        # K.layers.Dropout(dropout_rate): identity (Q6)
        layer = self.dense(layer)
        layer = torch.sigmoid(layer)
        layer = self.gather(layer)
        return self.out(layer)


class GIN(nn.Module):
    """example_model/model_gin.py:29-78."""

    def __init__(self, adj_channel_num=1, num_classes=2):
        super().__init__()
        self.agg = nn.ModuleList([layers.GINAggregate(adj_channel_num) for _ in range(2)])
        self.dense = nn.ModuleList([layers.GraphDense(50) for _ in range(4)])
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None):
        layer = features
        outs = []
        for blk in range(2):
            layer = self.agg[blk](layer, adj=adjs)
            layer = torch.relu(self.dense[2 * blk](layer))
            layer = torch.relu(self.dense[2 * blk + 1](layer))
            outs.append(layer)
        read_out = [self.gather(o) for o in outs]
        return self.out(torch.cat(read_out, dim=1))
