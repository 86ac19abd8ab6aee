"""CPU restatement (numpy, fp64) of the reference's two remaining named model files -- forward, loss
and hand-written backward.  TEST INFRASTRUCTURE ONLY (never imported by kgcn_amd/).
PARITY STATUS: unpinned at TensorFlow (see oracle/kgcn_oracle.py); gradients are checked by finite
differences in tests/test_oracle_model.py.

  multitask -- example_model/model_multitask.py:45-101 (BASELINE config 4):
      GraphConv(256) s GraphConv(256) s GraphDense(256) s GraphConv(50) GraphBatchNormalization s
      GraphDense(50) s GraphGather Dense(label_dim); s = sigmoid; loss = mask * sum_tasks mask_label *
      (weighted) sigmoid cross entropy; cost_opt = reduce_mean over the padded batch.
  sparse    -- example_model/sparse.py:45-134 (BASELINE config 3), block-diagonal batch of ONE:
      3 x [GraphConv(256) relu] GraphDense(256) GraphBatchNormalization relu, per-molecule sum (tf.scan
      :83-94), tanh, Dense(num_classes), loss = reduce_sum(sparse softmax cross entropy).
BatchNormalization in inference mode (quirk Q6): x * gamma / sqrt(1 + 1e-3) + beta.
"""
import numpy as np

from . import kgcn_oracle as K

BN_SCALE = 1.0 / np.sqrt(1.0 + 1e-3)


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def _conv_params(rng, din, dout, channels):
    return ([K.glorot_uniform(rng, din, dout).astype(np.float64) for _ in range(channels)],
            [np.zeros((1, dout)) for _ in range(channels)])


# -------------------------------------------------------------------------------------------------
# model_multitask.py
# -------------------------------------------------------------------------------------------------
def multitask_init(rng, in_dim, label_dim, channels=1, widths=(256, 256, 256, 50, 50)):
    p = {}
    p["w1"], p["b1"] = _conv_params(rng, in_dim, widths[0], channels)
    p["w2"], p["b2"] = _conv_params(rng, widths[0], widths[1], channels)
    p["k3"], p["c3"] = K.glorot_uniform(rng, widths[1], widths[2]).astype(np.float64), np.zeros(widths[2])
    p["w4"], p["b4"] = _conv_params(rng, widths[2], widths[3], channels)
    p["gamma"], p["beta"] = np.ones(widths[3]), np.zeros(widths[3])
    p["k5"], p["c5"] = K.glorot_uniform(rng, widths[3], widths[4]).astype(np.float64), np.zeros(widths[4])
    p["ok"], p["ob"] = K.glorot_uniform(rng, widths[4], label_dim).astype(np.float64), np.zeros(label_dim)
    return p


def sigmoid_ce(logits, labels, pos_weight=None):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x z + log(1+exp(-|x|));
    tf.nn.weighted_cross_entropy_with_logits: (1-z) x + (1 + (q-1) z) (log(1+exp(-|x|)) + max(-x,0))."""
    x, z = logits, labels
    sp = np.log1p(np.exp(-np.abs(x)))
    if pos_weight is None:
        return np.maximum(x, 0) - x * z + sp
    lw = 1 + (pos_weight - 1) * z
    return (1 - z) * x + lw * (sp + np.maximum(-x, 0))


def sigmoid_ce_grad(logits, labels, pos_weight=None):
    s = sigmoid(logits)
    if pos_weight is None:
        return s - labels
    lw = 1 + (pos_weight - 1) * labels
    return (1 - labels) - lw * (1 - s)


def multitask_forward(p, x, adjs, labels, mask, mask_label, enabled_node_nums, pos_weight=None):
    c = {}
    N = x.shape[1]
    c["valid"] = (np.arange(N)[None, :] < np.asarray(enabled_node_nums)[:, None]).astype(np.float64)[:, :, None]
    c["h1"] = K.graphconv_fwd_fast(x, adjs, p["w1"], p["b1"]); c["s1"] = sigmoid(c["h1"])
    c["h2"] = K.graphconv_fwd_fast(c["s1"], adjs, p["w2"], p["b2"]); c["s2"] = sigmoid(c["h2"])
    c["h3"] = K.graphdense_fwd(c["s2"], p["k3"], p["c3"]); c["s3"] = sigmoid(c["h3"])
    c["h4"] = K.graphconv_fwd_fast(c["s3"], adjs, p["w4"], p["b4"])
    c["bn"] = (c["h4"] * (p["gamma"] * BN_SCALE) + p["beta"]) * c["valid"]      # kgcn/layers.py:196-210
    c["s4"] = sigmoid(c["bn"])
    c["h5"] = K.graphdense_fwd(c["s4"], p["k5"], p["c5"]); c["s5"] = sigmoid(c["h5"])
    c["pool"] = K.gather_fwd(c["s5"])
    c["logits"] = c["pool"] @ p["ok"] + p["ob"]
    ce = sigmoid_ce(c["logits"], labels, pos_weight)
    cost = mask * (mask_label * ce).sum(axis=1)                                   # model_multitask.py:70-76
    c["cost_opt"], c["cost_sum"] = cost.mean(), cost.sum()
    c["prediction"] = sigmoid(c["logits"])
    return c


def multitask_backward(p, c, x, adjs, labels, mask, mask_label, pos_weight=None):
    B, N = x.shape[0], x.shape[1]
    g = {}
    dlogits = (mask / B)[:, None] * mask_label * sigmoid_ce_grad(c["logits"], labels, pos_weight)
    g["ok"], g["ob"] = c["pool"].T @ dlogits, dlogits.sum(axis=0)
    ds5 = K.gather_bwd(dlogits @ p["ok"].T, N)
    dh5 = ds5 * c["s5"] * (1 - c["s5"])
    ds4, g["k5"], g["c5"] = K.graphdense_bwd(c["s4"], p["k5"], dh5)
    dbn = ds4 * c["s4"] * (1 - c["s4"]) * c["valid"]
    g["gamma"] = (dbn * c["h4"] * BN_SCALE).sum(axis=(0, 1))
    g["beta"] = dbn.sum(axis=(0, 1))
    dh4 = dbn * (p["gamma"] * BN_SCALE)
    ds3, g["w4"], g["b4"] = K.graphconv_bwd_fast(c["s3"], adjs, p["w4"], dh4)
    dh3 = ds3 * c["s3"] * (1 - c["s3"])
    ds2, g["k3"], g["c3"] = K.graphdense_bwd(c["s2"], p["k3"], dh3)
    dh2 = ds2 * c["s2"] * (1 - c["s2"])
    ds1, g["w2"], g["b2"] = K.graphconv_bwd_fast(c["s1"], adjs, p["w2"], dh2)
    dh1 = ds1 * c["s1"] * (1 - c["s1"])
    g["dx"], g["w1"], g["b1"] = K.graphconv_bwd_fast(x, adjs, p["w1"], dh1)
    return g


# -------------------------------------------------------------------------------------------------
# sparse.py
# -------------------------------------------------------------------------------------------------
def sparse_init(rng, in_dim, num_classes, channels=1, out_dims=(256, 256, 256), dense_dim=256):
    p = {}
    d = in_dim
    for i, o in enumerate(out_dims, 1):
        p["w%d" % i], p["b%d" % i] = _conv_params(rng, d, o, channels)
        d = o
    p["dk"], p["dc"] = K.glorot_uniform(rng, d, dense_dim).astype(np.float64), np.zeros(dense_dim)
    p["gamma"], p["beta"] = np.ones(dense_dim), np.zeros(dense_dim)
    p["ok"], p["ob"] = K.glorot_uniform(rng, dense_dim, num_classes).astype(np.float64), np.zeros(num_classes)
    return p


def sparse_forward(p, net, channels, sizes, labels):
    """net [sumN, F]; channels = the block-diagonal COO list; the layers see a batch of ONE
    (tf.expand_dims(net, 0), adj = [channels], sparse.py:65-69)."""
    c = {}
    adjs = [channels]
    nconv = sum(1 for k in p if k.startswith("w"))
    h = np.asarray(net, np.float64)[None]
    c["in0"] = h
    for i in range(1, nconv + 1):
        c["h%d" % i] = K.graphconv_fwd(h, adjs, p["w%d" % i], p["b%d" % i])
        h = np.maximum(c["h%d" % i], 0)
        c["in%d" % i] = h
    c["hd"] = K.graphdense_fwd(h, p["dk"], p["dc"])
    c["bn"] = c["hd"] * (p["gamma"] * BN_SCALE) + p["beta"]
    c["r"] = np.maximum(c["bn"], 0)[0]
    c["pool"] = K.segment_sum_fwd(c["r"], sizes)
    c["t"] = np.tanh(c["pool"])
    c["logits"] = c["t"] @ p["ok"] + p["ob"]
    z = c["logits"] - c["logits"].max(axis=1, keepdims=True)
    c["logp"] = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    c["loss"] = -c["logp"][np.arange(len(labels)), labels].sum()                 # sparse.py:112-113
    c["probabilities"] = np.exp(c["logp"])
    return c


def sparse_backward(p, c, channels, sizes, labels):
    adjs = [channels]
    nconv = sum(1 for k in p if k.startswith("w"))
    g = {}
    dlogits = c["probabilities"].copy()
    dlogits[np.arange(len(labels)), labels] -= 1.0
    g["ok"], g["ob"] = c["t"].T @ dlogits, dlogits.sum(axis=0)
    dpool = (dlogits @ p["ok"].T) * (1 - c["t"] ** 2)
    dr = K.segment_sum_bwd(dpool, sizes)[None]
    dbn = dr * (c["bn"] > 0)
    g["gamma"] = (dbn * c["hd"] * BN_SCALE).sum(axis=(0, 1))
    g["beta"] = dbn.sum(axis=(0, 1))
    dh, g["dk"], g["dc"] = K.graphdense_bwd(c["in%d" % nconv], p["dk"], dbn * (p["gamma"] * BN_SCALE))
    for i in range(nconv, 0, -1):
        dh = dh * (c["h%d" % i] > 0)
        dh, g["w%d" % i], g["b%d" % i] = K.graphconv_bwd(c["in%d" % (i - 1)], adjs, p["w%d" % i], p["b%d" % i], dh)[:3]
    g["dnet"] = dh[0]
    return g


# -------------------------------------------------------------------------------------------------
# model_gin.py (example_model/model_gin.py:40-78): two blocks [GINAggregate - GraphDense - relu - GraphDense - relu], each block
# output read out by GraphGather, concat, Dense(classes), masked softmax cross entropy (mean over the padded batch, quirk Q5)
# -------------------------------------------------------------------------------------------------
def gin_init(rng, in_dim, width, classes=2, channels=1):
    p = {"eps": [rng.standard_normal(channels) * 0.3 for _ in range(2)]}
    d = in_dim
    for i in range(4):
        if i == 2:
            d = width
        p["k%d" % i] = K.glorot_uniform(rng, d, width).astype(np.float64)
        p["c%d" % i] = rng.standard_normal(width) * 0.1
        d = width
    p["ok"], p["ob"] = K.glorot_uniform(rng, 2 * width, classes).astype(np.float64), rng.standard_normal(classes) * 0.1
    return p


def _gin_block_diag(adjs, n):
    return [K.block_diag_csr(adjs, c, n, np.float64) for c in range(len(adjs[0]))]


def gin_forward(p, x, adjs, labels, mask):
    """TEST INFRASTRUCTURE.  layers.py:461-472 (default branch: eps_c x + A_c x, summed over the channels) on a block-diagonal CSR."""
    B, N, _ = x.shape
    A = _gin_block_diag(adjs, N)
    c = {"A": A}
    h = np.asarray(x, np.float64).reshape(B * N, -1)
    pools = []
    for blk in range(2):
        c["in%d" % blk] = h
        a = sum(p["eps"][blk][ch] * h + A[ch] @ h for ch in range(len(A)))
        c["a%d" % blk] = a
        z0 = a @ p["k%d" % (2 * blk)] + p["c%d" % (2 * blk)]
        r0 = np.maximum(z0, 0)
        z1 = r0 @ p["k%d" % (2 * blk + 1)] + p["c%d" % (2 * blk + 1)]
        r1 = np.maximum(z1, 0)
        c["z0_%d" % blk], c["r0_%d" % blk], c["z1_%d" % blk], c["r1_%d" % blk] = z0, r0, z1, r1
        pools.append(r1.reshape(B, N, -1).sum(axis=1))
        h = r1
    c["pool"] = np.concatenate(pools, axis=1)
    c["logits"] = c["pool"] @ p["ok"] + p["ob"]
    z = c["logits"] - c["logits"].max(axis=1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    cost = mask * -(labels * logp).sum(axis=1)
    c["softmax"] = np.exp(logp)
    c["cost_opt"], c["cost_sum"] = cost.mean(), cost.sum()
    return c


def gin_backward(p, c, x, adjs, labels, mask, relu_masks=None):
    """relu_masks: {(block, layer): bool array} taken from another implementation's activations (a pre-activation of 1e-9
    may flip its sign between fp32 and fp64); None = the oracle's own."""
    B, N, _ = x.shape
    A = c["A"]
    W = p["k0"].shape[1]
    g = {"eps": [np.zeros(len(A)), np.zeros(len(A))]}
    dlogits = (mask / B)[:, None] * (c["softmax"] * labels.sum(axis=1, keepdims=True) - labels)
    g["ok"], g["ob"] = c["pool"].T @ dlogits, dlogits.sum(axis=0)
    dpool = dlogits @ p["ok"].T
    dh = None                                            # gradient handed down from the block above
    for blk in (1, 0):
        dr1 = np.repeat(dpool[:, blk * W:(blk + 1) * W], N, axis=0)
        if dh is not None:
            dr1 = dr1 + dh
        m1 = (c["z1_%d" % blk] > 0) if relu_masks is None else relu_masks[(blk, 1)].reshape(B * N, -1)
        dz1 = dr1 * m1
        g["k%d" % (2 * blk + 1)], g["c%d" % (2 * blk + 1)] = c["r0_%d" % blk].T @ dz1, dz1.sum(axis=0)
        dr0 = dz1 @ p["k%d" % (2 * blk + 1)].T
        m0 = (c["z0_%d" % blk] > 0) if relu_masks is None else relu_masks[(blk, 0)].reshape(B * N, -1)
        dz0 = dr0 * m0
        g["k%d" % (2 * blk)], g["c%d" % (2 * blk)] = c["a%d" % blk].T @ dz0, dz0.sum(axis=0)
        da = dz0 @ p["k%d" % (2 * blk)].T
        hin = c["in%d" % blk]
        dh = 0
        for ch in range(len(A)):
            g["eps"][blk][ch] = (da * hin).sum()
            dh = dh + p["eps"][blk][ch] * da + A[ch].T @ da
    g["dx"] = np.asarray(dh).reshape(x.shape)
    return g
