"""Model definitions mirroring the reference's model plugins line for line in LAYER CALLS
(SURVEY 8f, row N1).  The layers of the hot path run in HIP kernels (kgcn_amd.layers); the elementwise
activation the reference writes as its own TF op after a layer (tf.sigmoid(layer), tf.nn.relu(layer)) is
handed to the layer as `activation=` and computed in the epilogue of the kernel that produces the tensor
(aggregation or GEMM) -- same function, one pass over HBM less per layer; where a BatchNormalization sits
between layer and activation it is one HIP elementwise kernel (ops.activation).  The loss is torch.

  GCN  -- example_model/model.py:41-61      GraphConv(50) x3, BN, GraphDense(50), GraphGather, Dense(2)
  GIN  -- example_model/model_gin.py:40-67  2 x [GINAggregate, GraphDense(50) x2], Gather x2, Dense(2)
  GATNet -- example_model/model_gat.py:40-62  3 x [GraphDense(50), GAT], Gather of blocks 2 and 3, Dense(2)
  MultitaskGCN -- example_model/model_multitask.py:45-101  GraphConv 256/256, GraphDense 256, GraphConv 50, BN,
                  GraphDense 50, Gather, Dense(label_dim); masked (weighted) sigmoid cross entropy
  SparseGCN    -- example_model/sparse.py:45-134  block-diagonal batch of one: 3 x [GraphConv(256) relu],
                  GraphDense(256), BN, relu, per-molecule sum, tanh, Dense(num_classes); summed sparse softmax CE

Keras learning-phase semantics (quirk Q6): the reference calls BatchNormalization / Dropout
without `training=`; under TF1 graph mode that is inference behaviour -- BN normalises with its
moving statistics (0, 1) and Dropout is the identity.  That is what is implemented here
(layers.GraphBatchNormalization).
"""
import math

import torch
from torch import nn

from . import layers, ops, ragged as _ragged


GraphBatchNormalization = layers.GraphBatchNormalization


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
    cost_opt = reduce_mean(cost) over the PADDED batch (quirk Q5); cost_sum = reduce_sum(cost).
    One HIP pass (csrc/train.hip: per-graph cost, d cost_sum / d logits, fixed-order sums)."""
    return ops.masked_softmax_ce(logits, labels, mask)


class GCN(nn.Module):
    """example_model/model.py:30-71."""

    def __init__(self, adj_channel_num=1, num_classes=2, ragged=False):
        """ragged: run on the valid node rows only when enabled_node_nums is given (kgcn_amd.ragged; same results as the
        padded formulation, which computes every layer on all max_node_num rows)."""
        super().__init__()
        self.ragged = bool(ragged)
        self.conv1 = layers.GraphConv(50, adj_channel_num, activation="sigmoid")    # :42-43 tf.sigmoid(layer)
        self.conv2 = layers.GraphConv(50, adj_channel_num, activation="sigmoid")    # :44-45
        self.conv3 = layers.GraphConv(50, adj_channel_num)
        self.bn = GraphBatchNormalization(activation="sigmoid")                     # :48-50 tf.sigmoid(bn(...)), one pass
        self.dense = layers.GraphDense(50, activation="sigmoid")                    # :52-53
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None):
        features, adjs, enabled_node_nums, rb = _ragged.enter(self.ragged, features, adjs, enabled_node_nums)
        if rb is None:
            adjs = layers._pack(adjs, features)         # list-of-lists feed: packed ONCE per forward, not per layer
            # small graphs (N <= 32, widths <= 64, one channel): the whole node-level body in one launch per direction
            pooled = layers.fused_stack([self.conv1, self.conv2, self.conv3, self.bn, self.dense], features, adjs,
                                        enabled_node_nums=enabled_node_nums, gather=True)
            if pooled is not None:
                return self.out(pooled)
        layer = self.conv1(features, adj=adjs)
        layer = self.conv2(layer, adj=adjs)
        layer = self.conv3(layer, adj=adjs)
        layer = self.bn(layer, max_node_num=features.shape[1], enabled_node_nums=enabled_node_nums)
        # K.layers.Dropout(dropout_rate): identity (Q6)
        layer = self.dense(layer)
        layer = self.gather(layer, ragged=rb)
        return self.out(layer)


_GIN_JOIN = __import__("os").environ.get("KGCN_GIN_JOIN") != "0"          # (development A/B: "0" = torch.cat of the read-outs)


class GIN(nn.Module):
    """example_model/model_gin.py:29-78."""

    def __init__(self, adj_channel_num=1, num_classes=2, width=50):
        """width: units of the four GraphDense layers (50 in the file; BASELINE config 5 quotes the layer at 256)."""
        super().__init__()
        self.agg = nn.ModuleList([layers.GINAggregate(adj_channel_num) for _ in range(2)])
        self.dense = nn.ModuleList([layers.GraphDense(width, activation="relu") for _ in range(4)])   # :45-54 tf.nn.relu
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None):
        adjs = layers._pack(adjs, features)
        layer = features
        outs = []
        # tf.concat of the two read-outs (model_gin.py:61): each is written into its column block of ONE buffer (and its gradient
        # read out of the column block of d buffer): no concatenation pass, no copies of the strided gradient blocks
        width = self.dense[1].output_dim
        joined = features.new_empty((features.shape[0], 2 * width)) if width % 4 == 0 and _GIN_JOIN else None
        for blk in range(2):
            # (block 0: the features need no gradient, so d epsilon is formed inside the dense layer's dX GEMM -- layers.gin_graph_dense)
            layer = layers.gin_graph_dense(self.agg[blk], self.dense[2 * blk], layer, adj=adjs)
            # the block output is read out (and, for block 0, passed on): d pooled joins the gradient inside the layer's dX GEMM
            layer, pooled = layers.graph_dense_gather(self.dense[2 * blk + 1], layer, join=joined, join_col=blk * width)
            outs.append(pooled)
        return self.out(torch.cat(outs, dim=1) if joined is None else ops.join_columns(joined, outs))


def masked_sigmoid_ce(logits, labels, mask, mask_label, pos_weight=None):
    """model_multitask.py:66-79: cost = mask * sum_tasks mask_label * (weighted) sigmoid cross entropy;
    cost_opt = reduce_mean over the padded batch, cost_sum = reduce_sum.  Formulas of
    tf.nn.sigmoid_cross_entropy_with_logits / tf.nn.weighted_cross_entropy_with_logits; one HIP pass (csrc/train.hip)."""
    return ops.masked_sigmoid_ce(logits, labels, mask, mask_label, pos_weight)


def sparse_softmax_ce_sum(logits, labels):
    """sparse.py:112-113: loss_to_minimize = reduce_sum(sparse_softmax_cross_entropy_with_logits)."""
    return ops.sparse_softmax_ce_sum(logits, labels)


class MultitaskGCN(nn.Module):
    """example_model/model_multitask.py:32-101."""

    def __init__(self, adj_channel_num=1, label_dim=12, ragged=False):
        """ragged: run on the valid node rows only when enabled_node_nums is given (kgcn_amd.ragged) -- the reference
        computes GraphConv / GraphDense on all max_node_num rows and only its BN on the valid ones (:58-60); the results
        are the same, the padded rows' constant contribution to GraphGather included."""
        super().__init__()
        self.ragged = bool(ragged)
        self.conv1 = layers.GraphConv(256, adj_channel_num, activation="sigmoid")   # :51-52
        self.conv2 = layers.GraphConv(256, adj_channel_num, activation="sigmoid")   # :53-54
        self.dense1 = layers.GraphDense(256, activation="sigmoid")                  # :55-56
        self.conv3 = layers.GraphConv(50, adj_channel_num)
        self.bn = layers.GraphBatchNormalization(activation="sigmoid")              # :58-60 tf.sigmoid(bn(...)), one pass
        self.dense2 = layers.GraphDense(50, activation="sigmoid")                   # :61-62
        self.gather = layers.GraphGather()
        self.out = KerasDense(label_dim)

    def forward(self, features, adjs, enabled_node_nums=None):
        features, adjs, enabled_node_nums, rb = _ragged.enter(self.ragged, features, adjs, enabled_node_nums)
        if rb is None:
            adjs = layers._pack(adjs, features)
        layer = self.conv1(features, adj=adjs)
        layer = self.conv2(layer, adj=adjs)
        layer = self.dense1(layer)
        layer = self.conv3(layer, adj=adjs)
        layer = self.bn(layer, max_node_num=features.shape[1], enabled_node_nums=enabled_node_nums)
        layer = self.dense2(layer)
        layer = self.gather(layer, ragged=rb)
        return self.out(layer)                      # prediction = sigmoid(logits)


def wants_augmented_features(model, n_features):
    """True when `model`'s first layer is a one-kernel-per-step consumer of [x | 1 | 0] feature rows: a GraphConv that takes the
    aggregate-first route (layers.py: din + 1 padded to 4 below its width), so that a static ragged batch should assemble its rows
    in that form (data_util.DeviceGraphDataset.static_ragged_batch(augmented_features=True))."""
    first = next((m for m in model.children() if isinstance(m, (layers.GraphConv, layers.GraphDense, layers.GINAggregate))), None)
    if not isinstance(first, layers.GraphConv) or not getattr(model, "ragged", False):
        return False
    dp = (int(n_features) + 1 + 3) // 4 * 4
    return bool(layers.aggregate_first and dp < first.output_dim and not (layers.enabled_bconv or layers.enabled_bspmm or
                                                                          layers.enabled_batched))


class SparseGCN(nn.Module):
    """example_model/sparse.py:45-134 (params of build(): out_dims [256,256,256], dense_dim 256,
    batch_normalize False, max_pool False; both optional layers are available as flags)."""

    def __init__(self, num_classes, adj_channel_num=1, out_dims=(256, 256, 256), dense_dim=256,
                 batch_normalize=False, max_pool=False):
        super().__init__()
        fuse = None if (batch_normalize or max_pool) else "relu"        # relu directly behind the layer: in its epilogue
        self.convs = nn.ModuleList([layers.GraphConv(o, adj_channel_num, activation=fuse) for o in out_dims])
        self.pools = nn.ModuleList([layers.GraphMaxPooling(adj_channel_num) for _ in out_dims]) if max_pool else None
        self.bns = nn.ModuleList([layers.GraphBatchNormalization() for _ in out_dims]) if batch_normalize else None
        self.dense = layers.GraphDense(dense_dim)
        self.bn = layers.GraphBatchNormalization(activation="relu")     # tf.nn.relu(bn(dense)), one pass
        self.out = KerasDense(num_classes)

    def forward(self, batch):
        """batch: kgcn_amd.data_util.BlockDiagonalBatch."""
        net = batch.features.unsqueeze(0)                       # tf.expand_dims(net, 0)
        for i, conv in enumerate(self.convs):
            net = conv(net, batch.adjacency)                    # positional adj, sparse.py:69
            if self.pools is not None:
                net = self.pools[i](net, batch.adjacency)
            if self.bns is not None:
                net = self.bns[i](net)
            if conv.activation is None:
                net = ops.activation(net, "relu")
        net = self.bn(self.dense(net))
        net = net.reshape(net.shape[1], net.shape[2])           # (a view: indexing [0] costs a zero fill + a copy in backward)
        # per-molecule node sum (:83-94) with tf.tanh (:95) in the aggregation's epilogue; its derivative rides in the adjoint
        net = ops.bconv(batch.segments_adjacency(), net, net.shape[1], activation="tanh")
        return self.out(net)                                    # probabilities = softmax(logits)


class DeepChemGCN(nn.Module):
    """example_model/model_deepchem.py:31-81: 4 x [GraphConv(64 / 128 / 128 / 64) -> relu -> GraphMaxPooling ->
    GraphBatchNormalization (valid rows) -> Dropout], GraphDense(64) -> sigmoid, GraphGather, Dense(num_classes).
    dropout_rate: the `dropout_rate` placeholder (0 = the evaluation feed; torch's dropout mask otherwise)."""

    def __init__(self, adj_channel_num=1, num_classes=2, widths=(64, 128, 128, 64)):
        super().__init__()
        self.conv = nn.ModuleList([layers.GraphConv(w, adj_channel_num, activation="relu") for w in widths])   # :44-45 tf.nn.relu(conv)
        self.pool = nn.ModuleList([layers.GraphMaxPooling(adj_channel_num) for _ in widths])
        self.bn = nn.ModuleList([GraphBatchNormalization() for _ in widths])
        self.dense = layers.GraphDense(64, activation="sigmoid")                                              # :75-76
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None, dropout_rate=0.0):
        adjs = layers._pack(adjs, features)
        layer = features
        for conv, pool, bn in zip(self.conv, self.pool, self.bn):
            layer = bn(pool(conv(layer, adj=adjs), adj=adjs), enabled_node_nums=enabled_node_nums)
            if dropout_rate:
                layer = torch.nn.functional.dropout(layer, p=float(dropout_rate), training=True)
        return self.out(self.gather(self.dense(layer)))


class NodeLabelGCN(nn.Module):
    """example_model/model_node_label.py:48-62 (features given): GraphConv(64) -> GraphBatchNormalization -> relu, twice, then
    GraphConv(num_classes): per-NODE logits [B, N, num_classes]."""

    def __init__(self, adj_channel_num=1, num_classes=2):
        super().__init__()
        self.conv = nn.ModuleList([layers.GraphConv(64, adj_channel_num), layers.GraphConv(64, adj_channel_num),
                                   layers.GraphConv(num_classes, adj_channel_num)])
        self.bn = nn.ModuleList([GraphBatchNormalization(activation="relu") for _ in range(2)])       # :52-55 tf.nn.relu(bn(...))

    def forward(self, features, adjs, enabled_node_nums=None):
        adjs = layers._pack(adjs, features)
        layer = features
        for conv, bn in zip(self.conv[:2], self.bn):
            layer = bn(conv(layer, adj=adjs), enabled_node_nums=enabled_node_nums)
        return self.conv[2](layer, adj=adjs)


def node_softmax_ce(logits, node_labels, mask):
    """model_node_label.py:64-70: cost[b] = mask[b] * mean over ALL N node rows of softmax_cross_entropy(node_label[b, n],
    logits[b, n]) (the mask_node_label placeholder is read but not used there); cost_opt = reduce_mean over the padded batch,
    cost_sum = reduce_sum.  -> (cost_opt, cost_sum)"""
    ce = -(node_labels * torch.log_softmax(logits, dim=2)).sum(dim=2)
    cost = mask * ce.mean(dim=1)
    return cost.mean(), cost.sum()


class GATNet(nn.Module):
    """example_model/model_gat.py:30-80."""

    def __init__(self, adj_channel_num=1, num_classes=2):
        super().__init__()
        self.dense = nn.ModuleList([layers.GraphDense(50) for _ in range(3)])
        self.gat = nn.ModuleList([layers.GAT(adj_channel_num) for _ in range(3)])
        self.gather = layers.GraphGather()
        self.out = KerasDense(num_classes)

    def forward(self, features, adjs, enabled_node_nums=None):
        adjs = layers._pack(adjs, features)
        layer = features
        block_out = []
        for i in range(3):
            layer = self.gat[i](self.dense[i](layer), adj=adjs)
            if i > 0:
                block_out.append(layer)
        return self.out(torch.cat([self.gather(o) for o in block_out], dim=1))
