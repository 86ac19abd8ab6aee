"""kGCN layer API on PyTorch-ROCm tensors, backed by the HIP kernels of libkgcn_hip.so.

Mirrors the reference's layer surface for the hot path (kgcn/layers.py): same class names,
constructor arguments, call conventions, attribute names (.w[i], .bias[i], .epsilon[i],
.output_dim, .adj_channel_num), the module-level dispatch flags enabled_batched / enabled_bspmm /
enabled_bconv and load_bspmm(args) (kgcn/layers.py:14-29; KNIME pokes the flags directly,
KNIME/GCN-K/py/gcn_infer.py:530-535).  Parameters are created lazily at the first call from the
input's feature dimension, like Keras' build() (kgcn/layers.py:48-62).

`adj` is the reference's adjs[b][ch] list-of-lists of COO matrices (tuples (indices, values,
dense_shape) or objects with those attributes) or a pre-packed kgcn_amd.BatchedAdjacency.
"""
import math

import torch
from torch import nn

from . import ops
from .batched_csr import BatchedAdjacency, BatchedCSR, as_batched_adjacency

allow_local_batch_statistics = False      # see GraphBatchNormalization.forward: learning phase 1 under data parallelism
enabled_batched = False
enabled_bspmm = False
enabled_bconv = False
# default branch only: evaluate A (X W + b) as (A [X | 1]) [W ; b] when the input is narrower than the output (same function,
# fewer bytes: kgcn/layers.py:112-113 aggregates dout columns, this din + 1).  False = always contract first.
aggregate_first = True


def load_bspmm(args):
    """kgcn/layers.py:19-29 -- choose the dispatch variant from CLI-style flags; precedence
    batched > bspmm > bconv.  The reference additionally requires ./batched.so|bspmm.so|bconv.so
    in the working directory; here the three op contracts are always available (they are entry
    points of libkgcn_hip.so), so only the flags decide."""
    global enabled_batched, enabled_bspmm, enabled_bconv
    enabled_batched = enabled_bspmm = enabled_bconv = False
    if getattr(args, "batched", False):
        enabled_batched = True
    elif getattr(args, "bspmm", False):
        enabled_bspmm = True
    elif getattr(args, "bconv", False):
        enabled_bconv = True


def _init_tensor(shape, initializer, device):
    """Keras initializer names used on this path: 'glorot_uniform' (kernels), 'zeros'."""
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if callable(initializer):
        with torch.no_grad():
            t.copy_(torch.as_tensor(initializer(shape), dtype=torch.float32))
    elif initializer == "glorot_uniform":
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (1, 1)
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        nn.init.uniform_(t, -lim, lim)
    elif initializer == "zeros":
        nn.init.zeros_(t)
    elif initializer == "ones":
        nn.init.ones_(t)
    else:
        raise ValueError("unsupported initializer %r" % (initializer,))
    return t


def _init_vector(d, initializer, device, fan=None):
    """1-D Keras kernels (GraphDecoderDistMult's [Din]; rows of DistMult's [C, Din]): glorot limits from the
    variable's own shape, as Keras computes them (fan_in = fan_out = d for a vector)."""
    t = torch.empty((d,), dtype=torch.float32, device=device)
    if initializer == "glorot_uniform":
        fi, fo = (d, d) if fan is None else fan
        lim = math.sqrt(6.0 / (fi + fo))
        nn.init.uniform_(t, -lim, lim)
    elif initializer == "zeros":
        nn.init.zeros_(t)
    elif initializer == "ones":
        nn.init.ones_(t)
    elif callable(initializer):
        with torch.no_grad():
            t.copy_(torch.as_tensor(initializer((d,)), dtype=torch.float32))
    else:
        raise ValueError("unsupported initializer %r" % (initializer,))
    return t


def _pack(adj, inputs):
    if adj is None:
        raise ValueError("adj is required")
    a = as_batched_adjacency(adj, n_nodes=int(inputs.shape[1]), device=inputs.device)
    if a.num_graphs != inputs.shape[0] or a.n_nodes != inputs.shape[1]:
        raise ValueError("adjacency batch (%d graphs x %d nodes) does not match inputs %s"
                         % (a.num_graphs, a.n_nodes, tuple(inputs.shape)))
    return a


class GraphConv(nn.Module):
    """kgcn/layers.py:32-119.  Out[b] = sum_c A[b][c] @ (X[b] @ kernel_c + bias_c)."""

    def __init__(self, output_dim, adj_channel_num, initializer="glorot_uniform", activation=None, **kwargs):
        """activation (extension, default None = the reference's layer): 'sigmoid' / 'relu' / 'tanh' -- the elementwise
        op the reference's MODELS apply to the layer output (example_model/model.py:43 tf.sigmoid(layer), sparse.py:76
        tf.nn.relu) -- computed in the epilogue of the aggregation kernel instead of a separate pass over HBM."""
        super().__init__()
        self.output_dim = int(output_dim)
        self.adj_channel_num = int(adj_channel_num)
        self.initializer = initializer
        self.activation = activation
        ops.act_code(activation)
        self.w = nn.ParameterList()
        self.bias = nn.ParameterList()
        self.built = False

    def build(self, input_shape, device=None):
        """kgcn/layers.py:48-62: kernel{i} [Din, Dout] (initializer), bias{i} [1, Dout] zeros."""
        din = int(input_shape[2])
        for _ in range(self.adj_channel_num):
            self.w.append(nn.Parameter(_init_tensor((din, self.output_dim), self.initializer, device)))
            self.bias.append(nn.Parameter(_init_tensor((1, self.output_dim), "zeros", device)))
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[1], self.output_dim

    def forward(self, inputs, adj=None):
        if not self.built:
            self.build(inputs.shape, inputs.device)
        a = _pack(adj, inputs)
        if a.num_channels != self.adj_channel_num:
            raise ValueError("layer has %d adjacency channels, adj has %d"
                             % (self.adj_channel_num, a.num_channels))
        B, N, din = inputs.shape
        C, dout = self.adj_channel_num, self.output_dim
        x2d = inputs.reshape(B * N, din)
        act = self.activation
        if a.values is not None or enabled_bconv or enabled_bspmm or enabled_batched:
            # the reference-structured branches keep the reference's op boundaries: activation as its own op
            return ops.activation(self._forward_reference_branches(inputs, a, x2d, B, N, din, C, dout), act)
        # default branch (kgcn/layers.py:105-116), MI355X form: the whole layer in one kernel when
        # the shape fits the fused kernel, else one GEMM with concatenated kernels + fused
        # multi-channel aggregation (one launch for all channels, activation in its epilogue).
        if C == 1 and ops.graphconv_fused_supported(a.channels[0], din, dout):
            return ops.activation(ops.graphconv_fused(inputs, self.w[0], self.bias[0], a.channels[0]), act)
        dp = (din + 1 + 3) // 4 * 4
        if aggregate_first and dp < dout and B * N >= 1024:
            # A (X W + 1 b) = (A [X | 1]) [W ; b]: the aggregation runs over din + 1 (padded to a multiple of 4) columns
            # instead of dout, the activation rides in the GEMM epilogue, and the backward needs NO dout-wide adjoint
            # aggregation (dW, db come out of ONE weight-gradient GEMM over [A X | rowsum(A)]; d inputs -- when asked for --
            # is a dp-wide adjoint).  Per channel the operand is aggregated with that channel's adjacency; the channels are
            # concatenated along the contraction axis.
            xa = getattr(inputs, "_kgcn_aug", None)     # rows assembled as [x | 1 | 0] already (ragged.StaticRaggedBatch)
            if xa is None or tuple(xa.shape) != (B * N, dp) or inputs.requires_grad or xa.data_ptr() != inputs.data_ptr():
                xa = ops.augment_ones(x2d, dp)
            if getattr(self, "_agg_pad", None) is None or tuple(self._agg_pad.shape) != (dp - din - 1, dout) or \
                    self._agg_pad.device != inputs.device:
                self._agg_pad = self.w[0].new_zeros((dp - din - 1, dout))          # constant: allocated and zeroed once
            pad = self._agg_pad
            if C == 1:
                z0 = ops.bspmm(a.channels[0], xa)
                if not z0.requires_grad:
                    # [W; b; 0] is never assembled: its fragment table is split from the two parameters by the step's one table launch
                    return ops.dense_stacked(z0, self.w[0], self.bias[0], activation=act).reshape(B, N, dout)
                return ops.dense(z0, ops.stack_rows(self.w[0], self.bias[0], pad), None, activation=act).reshape(B, N, dout)
            # several channels: [A_0 X' | A_1 X' | ...] from ONE launch that reads X' once, the stacked operand from one more
            z = ops.fan_out(a, xa)
            return ops.dense(z, ops.stack_channel_rows(self.w, self.bias, pad), None, activation=act).reshape(B, N, dout)
        if C == 1:
            fw = ops.dense(x2d, self.w[0], self.bias[0])
        else:
            fw = ops.dense(x2d, *ops.cat_channels(self.w, self.bias))         # [W_0 | W_1 | ...]: one launch, no torch.cat
        return ops.bconv(a, fw, dout, activation=act).reshape(B, N, dout)

    def _forward_reference_branches(self, inputs, a, x2d, B, N, din, C, dout):
        if a.values is not None:
            # adjacency values are differentiable inputs (integrated gradients over `adjs`,
            # kgcn/visualization.py:207-210): one Bspmm per channel with its d values gradient
            o = None
            for c in range(C):
                oo = ops.bspmm(a.channels[c], ops.dense(x2d, self.w[c], self.bias[c]), values=a.values[c])
                o = oo if o is None else o + oo
            return o.reshape(B, N, dout)
        if enabled_bconv:
            # kgcn/layers.py:68-78: FW[b][ch] for every channel, ONE fused op (SpMM + add-n)
            fw = ops.dense(x2d, *ops.cat_channels(self.w, self.bias))
            return ops.bconv(a, fw, dout).reshape(B, N, dout)
        if enabled_bspmm:
            # kgcn/layers.py:79-90: one Bspmm per channel, channel results added
            o = None
            for c in range(C):
                fw = ops.dense(x2d, self.w[c], self.bias[c])
                oo = ops.bspmm(a.channels[c], fw)
                o = oo if o is None else o + oo
            return o.reshape(B, N, dout)
        if enabled_batched:
            # kgcn/layers.py:91-104: one [B*N, Din] GEMM + Bspmdt per channel, reduce_sum
            o = [ops.bspmm(a.channels[c], ops.dense(x2d, self.w[c], self.bias[c])) for c in range(C)]
            return torch.stack(o).sum(0).reshape(B, N, dout) if C > 1 else o[0].reshape(B, N, dout)
        raise AssertionError("unreachable")


class GraphDense(nn.Module):
    """kgcn/layers.py:223-265: Keras Dense applied to every node row; kernel [Din, Dout]
    glorot-uniform, bias [Dout] zeros, no activation (Keras defaults)."""

    def __init__(self, output_dim, use_bias=True, kernel_initializer="glorot_uniform", activation=None, **kwargs):
        """activation (extension, default None): the elementwise op the models apply to the output
        (model_gin.py:47 tf.nn.relu(GraphDense(...)), model.py:53 tf.sigmoid) in the GEMM epilogue."""
        super().__init__()
        self.output_dim = int(output_dim)
        self.activation = activation
        ops.act_code(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.kernel = None
        self.bias = None
        self.built = False

    def build(self, input_shape, device=None):
        din = int(input_shape[2])
        self.kernel = nn.Parameter(_init_tensor((din, self.output_dim), self.kernel_initializer, device))
        if self.use_bias:
            self.bias = nn.Parameter(_init_tensor((self.output_dim,), "zeros", device))
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[1], self.output_dim

    def forward(self, inputs, enabled_node_nums=None, shape=None, max_node_num=None, **kwargs):
        if not self.built:
            self.build(inputs.shape, inputs.device)
        B, N, din = inputs.shape
        if enabled_node_nums is None:
            return ops.dense(inputs.reshape(B * N, din), self.kernel, self.bias,
                             activation=self.activation).reshape(B, N, self.output_dim)
        y = ops.dense(inputs.reshape(B * N, din), self.kernel, self.bias).reshape(B, N, self.output_dim)
        if enabled_node_nums is not None:
            # ragged path (kgcn/layers.py:243-254): rows >= enabled_node_nums[b] are zero padding
            en = torch.as_tensor(enabled_node_nums, device=inputs.device).reshape(B, 1)
            mask = (torch.arange(N, device=inputs.device).reshape(1, N) < en).to(y.dtype)
            y = y * mask.unsqueeze(-1)
        return ops.activation(y, self.activation)          # the model's activation follows the zero padding


def graph_dense_gather(dense_layer, inputs, join=None, join_col=0):
    """GraphDense (with its fused activation) + GraphGather as one op (ops.dense_gather): -> (layer output [B, N, D], pooled
    [B, D]).  The backward of a wide layer forms d pooled's broadcast inside its dX GEMM.  `dense_layer`: a GraphDense applied
    without enabled_node_nums.  join / join_col: the read-out goes into a column block of the buffer `join` (ops.join_columns)."""
    if not dense_layer.built:
        dense_layer.build(inputs.shape, inputs.device)
    return ops.dense_gather(inputs, dense_layer.kernel, dense_layer.bias, activation=dense_layer.activation, join=join,
                            join_col=join_col)


class GINAggregate(nn.Module):
    """kgcn/layers.py:400-475: Out[b] = sum_c (epsilon_c X[b] + A[b][c] @ X[b]).

    The reference's bconv/bspmm/batched branches (:429-460) silently drop the epsilon term
    (SURVEY quirk Q1); that behaviour is reproduced when one of the dispatch flags is set."""

    def __init__(self, adj_channel_num, initializer="zeros", **kwargs):
        super().__init__()
        self.adj_channel_num = int(adj_channel_num)
        self.initializer = initializer
        self.epsilon = nn.ParameterList()
        self.built = False

    def build(self, input_shape, device=None):
        for _ in range(self.adj_channel_num):
            self.epsilon.append(nn.Parameter(_init_tensor((), self.initializer, device)))
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape

    def forward(self, inputs, adj=None):
        if not self.built:
            self.build(inputs.shape, inputs.device)
        a = _pack(adj, inputs)
        if a.num_channels != self.adj_channel_num:
            raise ValueError("layer has %d adjacency channels, adj has %d"
                             % (self.adj_channel_num, a.num_channels))
        drop_eps = enabled_bconv or enabled_bspmm or enabled_batched
        # one channel: a view of the parameter (torch.stack is a copy launch forward and a slice launch backward)
        eps = None if drop_eps else (self.epsilon[0].reshape(1) if len(self.epsilon) == 1 else torch.stack(list(self.epsilon)))
        return ops.gin_aggregate(inputs, eps, a)


def gin_graph_dense(agg_layer, dense_layer, inputs, adj=None):
    """GINAggregate followed by GraphDense (with its fused activation) as one op (ops.gin_dense) -- example_model/model_gin.py:45-50.
    When the inputs need no gradient (a model's first block) d epsilon = <d out, x> is accumulated inside the dense layer's dX GEMM
    and the gradient of the aggregation's output never exists in HBM; in every other case this IS the two layers one after the other.
    `dense_layer`: a GraphDense applied without enabled_node_nums."""
    if not agg_layer.built:
        agg_layer.build(inputs.shape, inputs.device)
    if not dense_layer.built:
        dense_layer.build(inputs.shape, inputs.device)
    a = _pack(adj, inputs)
    if (enabled_bconv or enabled_bspmm or enabled_batched) or len(agg_layer.epsilon) != 1 or a.num_channels != 1:
        return dense_layer(agg_layer(inputs, adj=a))
    return ops.gin_dense(inputs, agg_layer.epsilon[0].reshape(1), a, dense_layer.kernel, dense_layer.bias,
                         activation=dense_layer.activation)


class GraphMaxPooling(nn.Module):
    """kgcn/layers.py:122-153: out[b,i,k] = sum_c max_j dense(A[b][c] .* X[b][:,k])[i,j] -- the maximum
    of a_ij * x_jk over node i's stored neighbours (0 is a candidate unless the row is full)."""

    def __init__(self, adj_channel_num, **kwargs):
        super().__init__()
        self.adj_channel_num = int(adj_channel_num)

    def compute_output_shape(self, input_shape):
        return input_shape

    def forward(self, inputs, adj=None):
        if hasattr(adj, "graph_ptr") and hasattr(adj, "adjacency"):
            raise ValueError("GraphMaxPooling is not defined on a ragged-compact batch (its implicit-zero rule counts the "
                             "padded columns of the row, kgcn/layers.py:139-147): use the padded layout")
        a = _pack(adj, inputs)
        if a.num_channels != self.adj_channel_num:
            raise ValueError("layer has %d adjacency channels, adj has %d"
                             % (self.adj_channel_num, a.num_channels))
        return ops.graph_maxpool(inputs, a)


class GAT(nn.Module):
    """kgcn/layers.py:477-542: graph attention WITHOUT its own weight matrix (the reference's note: put a
    GraphDense in front).  Per channel a vector weight_a{i} [2*Din, 1] (initializer, default glorot
    uniform); out = sum_c sigmoid(sum_e alpha_e x[col_e]) with the softmax-like alpha of the reference
    (denominator gathered at the column index, padded / empty rows give sigmoid(0) = 0.5)."""

    def __init__(self, adj_channel_num, initializer="glorot_uniform", input_dim=None, **kwargs):
        super().__init__()
        self.adj_channel_num = int(adj_channel_num)
        self.initializer = initializer
        self.input_dim = input_dim
        self.weight_a = nn.ParameterList()
        self.built = False

    def build(self, input_shape, device=None):
        d = int(input_shape[2] if self.input_dim is None else self.input_dim)
        for _ in range(self.adj_channel_num):
            self.weight_a.append(nn.Parameter(_init_tensor((2 * d, 1), self.initializer, device)))
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape

    def forward(self, inputs, adj=None):
        if not self.built:
            self.build(inputs.shape, inputs.device)
        a = _pack(adj, inputs)
        if a.num_channels != self.adj_channel_num:
            raise ValueError("layer has %d adjacency channels, adj has %d"
                             % (self.adj_channel_num, a.num_channels))
        return ops.gat(inputs, a, list(self.weight_a))


class GraphDecoderInnerProd(nn.Module):
    """kgcn/layers.py:268-282: adj_hat[b] = X[b] X[b]^T."""

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[1], input_shape[1]

    def forward(self, inputs, **kwargs):
        return ops.gram(inputs)


class GraphDecoderDistMult(nn.Module):
    """kgcn/layers.py:285-305: adj_hat[b] = (kernel * X[b]) X[b]^T, kernel [Din] (initializer)."""

    def __init__(self, initializer="glorot_uniform", **kwargs):
        super().__init__()
        self.initializer = initializer
        self.w = nn.ParameterList()
        self.built = False

    def build(self, input_shape, device=None):
        self.w.append(nn.Parameter(_init_vector(int(input_shape[2]), self.initializer, device)))
        self.built = True

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[1], input_shape[1]

    def forward(self, inputs, **kwargs):
        if not self.built:
            self.build(inputs.shape, inputs.device)
        return ops.gram(inputs, self.w[0])


class DistMult(nn.Module):
    """kgcn/layers.py:307-361: one relation vector per adjacency channel, kernel [C, Din];
    call -> [B, C, N, N]; compute_score / compute_left_prediction / compute_right_prediction as the reference."""

    def __init__(self, initializer="glorot_uniform", adj_channel_num=1, **kwargs):
        super().__init__()
        self.initializer = initializer
        self.adj_channel_num = int(adj_channel_num)
        self.w = nn.ParameterList()
        self.built = False

    def build(self, input_shape, device=None):
        d = int(input_shape[-1])
        t = torch.stack([_init_vector(d, self.initializer, device, fan=(self.adj_channel_num, d))
                         for _ in range(self.adj_channel_num)])
        self.w.append(nn.Parameter(t))
        self.built = True

    def _ensure(self, like):
        if not self.built:
            self.build(like.shape, like.device)

    def compute_score(self, layer1, layer2, channel, **kwargs):          # :321-325
        self._ensure(layer1)
        return (layer1 * layer2 * self.w[0][channel]).sum(dim=1)

    def compute_left_prediction(self, layer, right_layer, channel, **kwargs):   # :327-336
        self._ensure(layer)
        return ops.dense(right_layer * self.w[0][channel], layer.t().contiguous())

    def compute_right_prediction(self, left_layer, layer, channel, **kwargs):   # :338-347
        self._ensure(left_layer)
        return torch.matmul(layer, (left_layer * self.w[0][channel]).unsqueeze(2)).squeeze(2)

    def compute_output_shape(self, input_shape):
        return input_shape[0], self.adj_channel_num, input_shape[1], input_shape[1]

    def forward(self, inputs, **kwargs):
        self._ensure(inputs)
        return torch.stack([ops.gram(inputs, self.w[0][i]) for i in range(self.adj_channel_num)], dim=1)


class BatchGraphConv(nn.Module):
    """kgcn/layers.py:363-397: relu(A @ (X W + b)) on ONE block-diagonal sparse matrix, inputs = [net [sumN, Din],
    adj].  The reference's build loop overwrites self.w / self.bias per channel, so exactly one kernel / bias
    pair exists whatever adj_channel_num is; bias [Dout]."""

    def __init__(self, output_dim, adj_channel_num=1, initializer="glorot_uniform", input_dim=None, **kwargs):
        super().__init__()
        self.output_dim = int(output_dim)
        self.adj_channel_num = int(adj_channel_num)
        self.initializer = initializer
        self.input_dim = input_dim
        self.w = None
        self.bias = None

    def compute_output_shape(self, input_shape):
        return input_shape[0][0], self.output_dim

    def forward(self, inputs, **kwargs):
        net, adj = inputs[0], inputs[1]
        if self.w is None:
            din = int(net.shape[1] if self.input_dim is None else self.input_dim)
            self.w = nn.Parameter(_init_tensor((din, self.output_dim), self.initializer, net.device))
            self.bias = nn.Parameter(_init_tensor((self.output_dim,), "zeros", net.device))
        n = int(net.shape[0])
        if isinstance(adj, BatchedAdjacency):
            csr = adj.channels[0]
        elif isinstance(adj, BatchedCSR):
            csr = adj
        else:                                   # one COO matrix (tuple or SparseTensor-like), [sumN, sumN]
            csr = BatchedCSR.from_coo_list([adj], rows=n, cols=n, device=net.device)
        fw = ops.dense(net, self.w, self.bias)
        return torch.relu(ops.bspmm(csr, fw))


_learning_phase = 0


def set_learning_phase(value):
    """K.set_learning_phase: 0 = inference behaviour (the TF1 default the reference runs under),
    1 = training behaviour of the Keras layers wrapped by this module (GraphBatchNormalization)."""
    global _learning_phase
    if value not in (0, 1, False, True):
        raise ValueError("learning phase must be 0 or 1")
    _learning_phase = int(value)


def learning_phase():
    return _learning_phase


class GraphBatchNormalization(nn.Module):
    """kgcn/layers.py:170-220: tf.keras.layers.BatchNormalization (momentum 0.99, epsilon 1e-3, gamma ones, beta zeros,
    moving mean 0 / variance 1 -- Keras defaults, TF-internal) over the VALID node rows -- the first
    enabled_node_nums[b] rows of every graph (:196-210; all rows when it is None, :211-216) -- zeros on the padding rows.

    Mode (quirk Q6): the reference calls the Keras layer without `training=`, so the Keras LEARNING PHASE decides; its
    `training` keyword is handed to the constructor as `trainable` (:205, :215), i.e. it only freezes gamma / beta.
      phase 0 (default, what TF1 graph mode gives the reference): normalise with the moving statistics,
          y = gamma (x - moving_mean) / sqrt(moving_variance + eps) + beta;
      phase 1 (layers.set_learning_phase(1), or learning_phase=1 on the layer): batch statistics over the valid rows
          (population variance, as tf.nn.moments), moving <- moving * momentum + batch * (1 - momentum), and the
          backward differentiates through the statistics.
    Statistics, normalisation and backward are HIP kernels (kgcn_graph_bn_*_f32, csrc/bn.hip).
    activation (not in the reference's signature): the tf.sigmoid / tf.nn.relu the model files apply to the layer's output
    (example_model/model.py:50, model_multitask.py:60), fused into the normalise pass and the backward's reads."""

    def __init__(self, bn_name=None, eps=1e-3, momentum=0.99, learning_phase=None, activation=None, **kwargs):
        super().__init__()
        self.activation = activation
        self.bn_name = bn_name
        self.eps = eps
        self.momentum = momentum
        self.learning_phase = learning_phase        # None: follow layers.set_learning_phase()
        self.gamma = None
        self.beta = None

    def compute_output_shape(self, input_shape):
        return input_shape

    def build(self, input_shape, device=None):
        d = int(input_shape[-1])
        dev = device if device is not None else "cuda"
        self.gamma = nn.Parameter(torch.ones(d, device=dev))
        self.beta = nn.Parameter(torch.zeros(d, device=dev))
        self.register_buffer("moving_mean", torch.zeros(d, device=dev))
        self.register_buffer("moving_variance", torch.ones(d, device=dev))

    def forward(self, x, enabled_node_nums=None, shape=None, max_node_num=None, training=True):
        if self.gamma is None:
            self.build(x.shape, x.device)
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        en = None
        if enabled_node_nums is not None:
            en = torch.as_tensor(enabled_node_nums, device=x.device).to(torch.int32).reshape(-1).contiguous()
            if en.numel() != x.shape[0]:
                raise ValueError("enabled_node_nums has %d entries for a batch of %d graphs" % (en.numel(), x.shape[0]))
        gamma, beta = (self.gamma, self.beta) if training else (self.gamma.detach(), self.beta.detach())
        phase = _learning_phase if self.learning_phase is None else int(self.learning_phase)
        if phase:
            # Batch statistics are statistics of the GLOBAL batch; under data parallelism each rank sees its shard only.  The
            # named model files run this layer in learning phase 0 (quirk Q6), so the BASELINE configurations never get here; a
            # phase-1 model on several ranks would silently diverge from the single-process result (the backward through the
            # statistics needs the same exchange) -- refused rather than approximated.
            import torch.distributed as _dist
            if _dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1 and not allow_local_batch_statistics:
                raise RuntimeError("GraphBatchNormalization with batch statistics (learning phase 1) under %d data-parallel ranks: "
                                   "the statistics of a rank's shard are not those of the global batch.  Run the layer in learning "
                                   "phase 0 (what the reference's TF1 graph mode does), or set kgcn_amd.layers."
                                   "allow_local_batch_statistics = True to accept per-rank statistics knowingly" % _dist.get_world_size())
            mean, var = ops.graph_bn_stats(x.detach(), en)
            with torch.no_grad():
                self.moving_mean.mul_(self.momentum).add_(mean, alpha=1.0 - self.momentum)
                self.moving_variance.mul_(self.momentum).add_(var, alpha=1.0 - self.momentum)
            y = ops.graph_bn(x, gamma, beta, mean, var, en, self.eps, True, self.activation)
        else:
            y = ops.graph_bn(x, gamma, beta, self.moving_mean, self.moving_variance, en, self.eps, False, self.activation)
        return y[0] if squeeze else y


class GraphGather(nn.Module):
    """kgcn/layers.py:156-167: reduce_sum over the node axis (padding rows included, quirk Q4)."""

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[2]

    def forward(self, inputs, ragged=None, **kwargs):
        """ragged: the kgcn_amd.ragged.RaggedBatch `inputs` [1, capacity, D] lives on -- per-graph sums over graph_ptr plus
        the padded rows' share (N - n_b times the padding representative row), i.e. the padded formulation's result."""
        if ragged is not None:
            return ops.ragged_gather(inputs, ragged)
        return ops.graph_gather(inputs)


# default on: models run eligible layer sequences through the cross-layer kernels (see fused_stack) for batches of at most
# stack_fusion_max_rows node rows (graphs x nodes): one launch per direction wins where a step is bound by launch latency (up
# to a few hundred graphs: one graph per workgroup trip, plain fp32 FMAs; thousands: 64-row tiles on the f32 MFMA -- 4,096
# graphs of 10 nodes still 5 % ahead); above that the per-layer bf16-split kernels are faster (tools/stack_sweep.py)
stack_fusion = True
stack_fusion_max_rows = 49152


def fused_stack(seq, features, adj, enabled_node_nums=None, gather=True):
    """Run the layer sequence `seq` -- GraphConv (one channel) / GraphBatchNormalization (learning phase 0) / GraphDense
    modules, each with its fused `activation` -- followed by GraphGather (gather=True) through the cross-layer kernels of
    csrc/stack.hip: ONE forward and ONE backward launch for the whole node-level body of example_model/model.py:42-54.
    Returns None when the sequence is not eligible (graphs of more than 32 nodes, widths above 64, several adjacency
    channels, differentiable adjacency values, a reference dispatch flag, BatchNormalization in its training phase, ragged
    GraphDense); the caller then runs the layers one by one.  Same function either way (tests/test_gpu_model.py)."""
    if not stack_fusion or enabled_batched or enabled_bspmm or enabled_bconv or features.dim() != 3:
        return None
    if features.shape[0] * features.shape[1] > stack_fusion_max_rows:
        return None
    if hasattr(adj, "graph_ptr") and hasattr(adj, "adjacency"):
        return None                                          # ragged-compact batches have their own route
    for m in seq:                                            # Keras build semantics: parameters exist after the first call
        if isinstance(m, (GraphConv, GraphDense)) and not m.built:
            return None
        if isinstance(m, GraphBatchNormalization) and m.gamma is None:
            return None
    a = _pack(adj, features)
    if a.num_channels != 1 or a.values is not None:
        return None
    csr = a.channels[0]
    spec, params, buffers = [], [], []
    d = int(features.shape[2])
    for m in seq:
        act = ops.act_code(m.activation)
        if isinstance(m, GraphConv):
            if m.adj_channel_num != 1 or m.w[0].shape[0] != d:
                return None
            spec.append((0, act, d, m.output_dim, 0.0)); params += [m.w[0], m.bias[0]]; buffers.append(None)
            d = m.output_dim
        elif isinstance(m, GraphDense):
            if m.kernel.shape[0] != d:
                return None
            spec.append((1, act, d, m.output_dim, 0.0)); params += [m.kernel, m.bias]; buffers.append(None)
            d = m.output_dim
        elif isinstance(m, GraphBatchNormalization):
            phase = _learning_phase if m.learning_phase is None else int(m.learning_phase)
            if phase or m.gamma.shape[0] != d:
                return None
            spec.append((2, act, d, d, float(m.eps))); params += [m.gamma, m.beta]
            buffers.append((m.moving_mean, m.moving_variance))
        else:
            return None
    if not ops.gcn_stack_supported(csr, spec):
        return None
    en = None
    if enabled_node_nums is not None and any(s[0] == 2 for s in spec):
        en = torch.as_tensor(enabled_node_nums, device=features.device).to(torch.int32).reshape(-1).contiguous()
        if en.numel() != features.shape[0]:
            raise ValueError("enabled_node_nums has %d entries for a batch of %d graphs" % (en.numel(), features.shape[0]))
    return ops.gcn_stack(features, csr, en, spec, buffers, gather, params)


__all__ = ["GraphConv", "GraphDense", "GINAggregate", "GraphGather", "GraphMaxPooling",
           "GraphBatchNormalization", "set_learning_phase", "learning_phase", "GAT", "GraphDecoderInnerProd",
           "GraphDecoderDistMult", "DistMult", "BatchGraphConv", "load_bspmm",
           "BatchedAdjacency", "fused_stack"]
