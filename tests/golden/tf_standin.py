"""A numpy stand-in for the ~30 TensorFlow / Keras symbols that /root/reference/kgcn/layers.py touches, so that the
reference's own LAYER TEXT (GraphConv.call, GraphDense.call, GINAggregate.call, GraphGather, GraphMaxPooling, GAT,
GraphBatchNormalization) can be executed in the build container, where TensorFlow cannot be installed.

What this is: test tooling for tests/golden/make_golden_layers.py.  It never travels to the GPU box, nothing of it is
imported by the product (kgcn_amd/) or by any test; only the arrays the reference's code returns over it are stored
(tests/golden/g6_*.npz).
What this is NOT: TensorFlow.  A stand-in for a missing library pins nothing -- the arithmetic of the reference stays
"parity unpinned" (DESIGN.md 4).  What the fixtures made over it do show is that the oracle's reading of the
reference's *Python* -- operation order, where the bias enters (quirk Q2), where epsilon enters (Q1), what GraphGather sums
(Q4), how the ragged GraphDense pads -- is what that Python computes when its primitives mean what their names say.
Every primitive below computes in float64 (the question is the formula, not TF's rounding) and documents the TF semantics it
assumes.
"""
import sys
import types

import numpy as np


class T(np.ndarray):
    """ndarray with the two Tensor methods the reference calls (set_shape; .shape is already a tuple of ints)."""

    def set_shape(self, shape):       # tf.Tensor.set_shape: static-shape annotation only
        assert tuple(int(s) for s in shape) == tuple(self.shape), (shape, self.shape)


def t(a):
    return np.asarray(a, dtype=np.float64).view(T)


class SparseTensor:
    """tf.SparseTensor(indices [nnz, 2], values [nnz], dense_shape [2])."""

    def __init__(self, indices, values, dense_shape):
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
        self.values = np.asarray(values, dtype=np.float64).reshape(-1)
        self.dense_shape = tuple(int(s) for s in dense_shape)

    def __mul__(self, dense):
        # tf sparse * dense: the dense operand is broadcast to the sparse shape, result keeps the sparse pattern
        # (kgcn/layers.py:141: adj_mat * inputs[b, :, k] -- a length-N vector broadcasts along the LAST axis: value * fb[col])
        d = np.broadcast_to(np.asarray(dense, dtype=np.float64), self.dense_shape)
        return SparseTensor(self.indices, self.values * d[self.indices[:, 0], self.indices[:, 1]], self.dense_shape)


def sparse_tensor_dense_matmul(sp, b):
    # tf.sparse_tensor_dense_matmul: out[row] += value * b[col] over the stored entries (duplicates accumulate)
    out = np.zeros((sp.dense_shape[0], b.shape[1]))
    np.add.at(out, sp.indices[:, 0], sp.values[:, None] * np.asarray(b)[sp.indices[:, 1]])
    return t(out)


def sparse_tensor_to_dense(sp):
    # tf.sparse_tensor_to_dense (validate_indices: no duplicates; absent entries are 0)
    out = np.zeros(sp.dense_shape)
    out[sp.indices[:, 0], sp.indices[:, 1]] = sp.values
    return t(out)


class _Initializers:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def make(self, name, shape):
        shape = tuple(int(s) for s in shape)
        if name == "zeros":
            return np.zeros(shape)
        if name == "ones":
            return np.ones(shape)
        if name == "glorot_uniform":      # Keras: limit sqrt(6 / (fan_in + fan_out))
            fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (1, 1)
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            return self.rng.uniform(-lim, lim, size=shape)
        raise NotImplementedError(name)


INIT = _Initializers(20260928)


class Layer:
    """tensorflow.python.keras.layers.Layer: build(input_shape) on first call, then call(...)."""

    def __init__(self, **kwargs):
        self.built = False
        self.weights_by_name = {}

    def add_weight(self, name=None, shape=(), initializer="zeros", trainable=True, **kw):
        w = t(INIT.make(initializer, shape))
        self.weights_by_name[name] = w
        return w

    def build(self, input_shape):
        self.built = True

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            shape = inputs[0].shape if isinstance(inputs, (list, tuple)) else inputs.shape
            self.build(tuple(shape))
            self.built = True
        return self.call(inputs, *args, **kwargs)


class Dense(Layer):
    """Keras Dense with default arguments: kernel glorot_uniform [in, units], bias zeros [units], no activation."""

    def __init__(self, units, **kwargs):
        self.units = int(units)
        super().__init__(**kwargs)

    def build(self, input_shape):
        self.kernel = self.add_weight(name="kernel", shape=(int(input_shape[-1]), self.units), initializer="glorot_uniform")
        self.bias = self.add_weight(name="bias", shape=(self.units,), initializer="zeros")
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        return t(np.asarray(inputs) @ np.asarray(self.kernel) + np.asarray(self.bias))


class BatchNormalization(Layer):
    """Keras BatchNormalization as a TF1 graph-mode call WITHOUT training= resolves it (quirk Q6: learning phase 0):
    moving statistics (0, 1) at initialisation, epsilon 1e-3, gamma 1, beta 0."""

    def __init__(self, trainable=True, name=None, epsilon=1e-3, **kwargs):
        self.epsilon = epsilon
        super().__init__(**kwargs)

    def build(self, input_shape):
        d = int(input_shape[-1])
        self.gamma, self.beta = self.add_weight("gamma", (d,), "ones"), self.add_weight("beta", (d,), "zeros")
        self.moving_mean, self.moving_variance = self.add_weight("moving_mean", (d,), "zeros"), self.add_weight("moving_variance", (d,), "ones")
        super().build(input_shape)

    def call(self, inputs, **kwargs):
        x = np.asarray(inputs)
        return t((x - self.moving_mean) / np.sqrt(np.asarray(self.moving_variance) + self.epsilon) * self.gamma + self.beta)


def install():
    """Put the stand-in modules into sys.modules under the names kgcn/layers.py imports."""
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "1.15.0"
    tf.SparseTensor = SparseTensor
    tf.SparseTensorValue = SparseTensor
    tf.matmul = lambda a, b: t(np.matmul(np.asarray(a), np.asarray(b)))
    tf.add = lambda a, b: t(np.asarray(a) + np.asarray(b))
    tf.add_n = lambda xs: t(sum(np.asarray(x) for x in xs))
    tf.stack = lambda xs, axis=0: t(np.stack([np.asarray(x) for x in xs], axis=axis))
    tf.unstack = lambda x, axis=0: [t(v) if np.ndim(v) else v for v in np.moveaxis(np.asarray(x), axis, 0)]
    tf.reshape = lambda x, shape: t(np.reshape(np.asarray(x), tuple(int(s) for s in shape)))
    tf.reduce_sum = lambda x, axis=None: t(np.sum(np.asarray(x), axis=axis))
    tf.reduce_max = lambda x, axis=None: t(np.max(np.asarray(x), axis=axis))
    tf.concat = lambda xs, axis: t(np.concatenate([np.asarray(x) for x in xs], axis=axis))
    tf.split = lambda x, sizes, axis=0: [t(p) for p in np.split(np.asarray(x), np.cumsum(np.asarray(sizes, dtype=np.int64))[:-1], axis=axis)]
    tf.pad = lambda x, paddings: t(np.pad(np.asarray(x), [(int(a), int(b)) for a, b in paddings]))
    tf.shape = lambda x: np.asarray(np.shape(x), dtype=np.int64)
    tf.transpose = lambda x, perm=None: t(np.transpose(np.asarray(x), perm))
    tf.gather = lambda x, idx, axis=0: t(np.take(np.asarray(x), np.asarray(idx, dtype=np.int64), axis=axis))
    tf.one_hot = lambda idx, depth: t(np.eye(int(depth))[np.asarray(idx, dtype=np.int64)])
    tf.exp = lambda x: t(np.exp(np.asarray(x)))
    tf.sigmoid = lambda x: t(1.0 / (1.0 + np.exp(-np.asarray(x))))
    tf.expand_dims = lambda x, axis: t(np.expand_dims(np.asarray(x), axis))
    tf.squeeze = lambda x, axis=None: t(np.squeeze(np.asarray(x), axis=tuple(axis) if axis is not None else None))
    tf.sparse_tensor_dense_matmul = sparse_tensor_dense_matmul
    tf.sparse_tensor_to_dense = sparse_tensor_to_dense
    nn = types.ModuleType("tensorflow.nn")
    nn.sigmoid = tf.sigmoid
    nn.relu = lambda x: t(np.maximum(np.asarray(x), 0.0))
    nn.leaky_relu = lambda x, alpha=0.2: t(np.where(np.asarray(x) > 0, np.asarray(x), alpha * np.asarray(x)))   # TF default alpha 0.2
    nn.bias_add = lambda x, b: t(np.asarray(x) + np.asarray(b))
    tf.nn = nn
    keras = types.ModuleType("tensorflow.keras")
    keras.layers = types.ModuleType("tensorflow.keras.layers")
    keras.layers.BatchNormalization = BatchNormalization
    keras.layers.Dense = Dense
    keras.layers.Layer = Layer
    tf.keras = keras
    py = types.ModuleType("tensorflow.python")
    pyk = types.ModuleType("tensorflow.python.keras")
    pykl = types.ModuleType("tensorflow.python.keras.layers")
    pykl.Layer, pykl.Dense = Layer, Dense
    py.keras, pyk.layers = pyk, pykl
    tf.python = py
    for name, mod in (("tensorflow", tf), ("tensorflow.nn", nn), ("tensorflow.keras", keras), ("tensorflow.keras.layers", keras.layers),
                      ("tensorflow.python", py), ("tensorflow.python.keras", pyk), ("tensorflow.python.keras.layers", pykl)):
        sys.modules[name] = mod
    return tf
