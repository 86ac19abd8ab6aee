"""CPU restatement (numpy, fp64) of the reference's example_model/model.py network, its loss and
tf.train.AdamOptimizer -- forward, hand-written backward, one train step.  TEST INFRASTRUCTURE ONLY.
PARITY STATUS: unpinned at TensorFlow (see oracle/kgcn_oracle.py); gradients are checked by finite
differences in tests/test_oracle_model.py.

Network (example_model/model.py:41-61):  GraphConv(50) - sigmoid - GraphConv(50) - sigmoid -
GraphConv(50) - GraphBatchNormalization (inference mode: x / sqrt(1 + 1e-3), quirk Q6) - sigmoid -
[Dropout = identity] - GraphDense(50) - sigmoid - GraphGather - Dense(2) - masked softmax CE with
reduce_mean over the padded batch (quirk Q5).  Optimiser: kgcn/core.py:121-127.
"""
import numpy as np

from . import kgcn_oracle as K

BN_SCALE = 1.0 / np.sqrt(1.0 + 1e-3)


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def init_params(rng, in_dim, channels=1, hidden=50, classes=2):
    p = {}
    d = in_dim
    for i in (1, 2, 3):
        p["w%d" % i] = [K.glorot_uniform(rng, d, hidden).astype(np.float64) for _ in range(channels)]
        p["b%d" % i] = [np.zeros((1, hidden)) for _ in range(channels)]
        d = hidden
    p["gamma"], p["beta"] = np.ones(hidden), np.zeros(hidden)
    p["dk"], p["db"] = K.glorot_uniform(rng, hidden, hidden).astype(np.float64), np.zeros(hidden)
    p["ok"], p["ob"] = K.glorot_uniform(rng, hidden, classes).astype(np.float64), np.zeros(classes)
    return p


def forward(p, x, adjs, labels, mask):
    c = {}
    c["h1"] = K.graphconv_fwd_fast(x, adjs, p["w1"], p["b1"])
    c["s1"] = sigmoid(c["h1"])
    c["h2"] = K.graphconv_fwd_fast(c["s1"], adjs, p["w2"], p["b2"])
    c["s2"] = sigmoid(c["h2"])
    c["h3"] = K.graphconv_fwd_fast(c["s2"], adjs, p["w3"], p["b3"])
    c["bn"] = c["h3"] * (p["gamma"] * BN_SCALE) + p["beta"]
    c["s3"] = sigmoid(c["bn"])
    c["h4"] = K.graphdense_fwd(c["s3"], p["dk"], p["db"])
    c["s4"] = sigmoid(c["h4"])
    c["pool"] = K.gather_fwd(c["s4"])
    c["logits"] = c["pool"] @ p["ok"] + p["ob"]
    z = c["logits"] - c["logits"].max(axis=1, keepdims=True)
    c["logp"] = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    cost = mask * -(labels * c["logp"]).sum(axis=1)
    c["cost_opt"], c["cost_sum"] = cost.mean(), cost.sum()
    return c


def backward(p, c, x, adjs, labels, mask):
    B, N = x.shape[0], x.shape[1]
    g = {}
    soft = np.exp(c["logp"])
    dlogits = (mask / B)[:, None] * (soft * labels.sum(axis=1, keepdims=True) - labels)
    g["ok"], g["ob"] = c["pool"].T @ dlogits, dlogits.sum(axis=0)
    dpool = dlogits @ p["ok"].T
    ds4 = K.gather_bwd(dpool, N)
    dh4 = ds4 * c["s4"] * (1 - c["s4"])
    ds3, g["dk"], g["db"] = K.graphdense_bwd(c["s3"], p["dk"], dh4)
    dbn = ds3 * c["s3"] * (1 - c["s3"])
    g["gamma"] = (dbn * c["h3"] * BN_SCALE).sum(axis=(0, 1))
    g["beta"] = dbn.sum(axis=(0, 1))
    dh3 = dbn * (p["gamma"] * BN_SCALE)
    ds2, g["w3"], g["b3"] = K.graphconv_bwd_fast(c["s2"], adjs, p["w3"], dh3)
    dh2 = ds2 * c["s2"] * (1 - c["s2"])
    ds1, g["w2"], g["b2"] = K.graphconv_bwd_fast(c["s1"], adjs, p["w2"], dh2)
    dh1 = ds1 * c["s1"] * (1 - c["s1"])
    _, g["w1"], g["b1"] = K.graphconv_bwd_fast(x, adjs, p["w1"], dh1)
    return g


class TFAdam:
    """tf.train.AdamOptimizer: lr_t = lr sqrt(1-b2^t)/(1-b1^t); p -= lr_t m / (sqrt(v) + eps)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m, self.v = {}, {}

    def _upd(self, key, p, g):
        m = self.m.get(key, 0.0) * self.b1 + (1 - self.b1) * g
        v = self.v.get(key, 0.0) * self.b2 + (1 - self.b2) * g * g
        self.m[key], self.v[key] = m, v
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        return p - lr_t * m / (np.sqrt(v) + self.eps)

    def step(self, p, g):
        self.t += 1
        for k in p:
            if isinstance(p[k], list):
                p[k] = [self._upd((k, i), a, np.asarray(b).reshape(a.shape)) for i, (a, b) in enumerate(zip(p[k], g[k]))]
            else:
                p[k] = self._upd(k, p[k], np.asarray(g[k]).reshape(p[k].shape))
        return p


def train_step(p, opt, x, adjs, labels, mask):
    c = forward(p, x, adjs, labels, mask)
    g = backward(p, c, x, adjs, labels, mask)
    return opt.step(p, g), c
