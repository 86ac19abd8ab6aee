#!/usr/bin/env python3
"""g6_*.npz: outputs of the REFERENCE'S OWN layer text (/root/reference/kgcn/layers.py, imported in place) executed over the
numpy stand-in of tests/golden/tf_standin.py -- GraphConv.call default branch (layers.py:105-116), GraphDense.call both paths
(:243-262), GINAggregate.call default (:461-472), GraphGather (:163-164), GraphMaxPooling (:135-150), GAT.call (:511-538),
GraphBatchNormalization.call both paths (:186-220) -- on batches fed by the reference's own loaders (kgcn/data_util.py,
kgcn/feed.py: the G3 batch of example_jbl/synthetic.jbl with 10 real + 20 dummy graphs; plain, Kipf-normalised and
degree-split adjacency) with seeded weights, NON-ZERO biases and epsilons.

Build container only (needs /root/reference); nothing of the reference is copied, only the arrays its code returns are stored.
Honest label: reference Python over stand-in primitives, float64.  Evidence that the oracle restates the reference's Python
(quirks Q1, Q2, Q4; the ragged GraphDense padding) -- NOT a pin of TensorFlow's arithmetic: parity stays "partial".

    python tests/golden/make_golden_layers.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import tf_standin  # noqa: E402


def flat_adjs(adjs):
    """adjs[b][ch] SparseTensor -> flat arrays."""
    idx, val, off = [], [], [0]
    for row in adjs:
        for a in row:
            idx.append(a.indices.astype(np.int32)); val.append(a.values); off.append(off[-1] + len(a.values))
    return dict(adj_idx=np.concatenate(idx).reshape(-1, 2), adj_val=np.concatenate(val), adj_off=np.asarray(off, np.int64),
                adj_shape=np.asarray(adjs[0][0].dense_shape, np.int64), adj_channels=np.int64(len(adjs[0])))


def main():
    tf_standin.install()
    sys.path.insert(0, REF)
    from kgcn import data_util, feed, layers        # reference code, imported in place
    rng = np.random.default_rng(6)
    out = {}
    for tag, norm, split in (("plain", False, False), ("norm", True, False), ("split_norm", True, True)):
        cfg = {"with_feature": True, "with_node_embedding": False, "normalize_adj_flag": norm, "split_adj_flag": split,
               "order": 1, "shuffle_data": False, "task": "multitask_classification"}
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            all_data, info = data_util.load_data(cfg, os.path.join(REF, "example_jbl/synthetic.jbl"), prohibit_shuffle=True)
        C, B = info.adj_channel_num, 30
        ph = {"adjs": [[("adj", ch, b) for ch in range(C)] for b in range(B)], "features": "features", "labels": "labels",
              "mask": "mask", "enabled_node_nums": "enabled_node_nums"}
        fd = feed.construct_feed(list(range(150, 160)), ph, all_data, batch_size=B, info=info, config=cfg)
        adjs = [[fd[("adj", ch, b)] for ch in range(C)] for b in range(B)]
        x0 = np.asarray(fd["features"], np.float64)
        # widen the 3 one-hot features with seeded noise so that no column is degenerate
        x = tf_standin.t(np.concatenate([x0, rng.standard_normal((B, x0.shape[1], 4)) * (x0.sum(2, keepdims=True) > 0)], axis=2))
        res = dict(x=np.asarray(x), **flat_adjs(adjs))
        # --- GraphConv (default branch), non-zero biases: the bias enters BEFORE the aggregation (quirk Q2)
        conv = layers.GraphConv(9, C)
        conv(x, adj=adjs)                                     # build
        for c in range(C):
            conv.bias[c][...] = rng.standard_normal(conv.bias[c].shape) * 0.3
        res["conv_out"] = np.asarray(conv(x, adj=adjs))
        res["conv_w"] = np.stack([np.asarray(w) for w in conv.w]); res["conv_b"] = np.stack([np.asarray(b) for b in conv.bias])
        # --- GINAggregate (default branch), non-zero epsilon (quirk Q1: the default branch keeps eps * x)
        gin = layers.GINAggregate(C)
        gin(x, adj=adjs)
        for c in range(C):
            gin.epsilon[c][...] = rng.standard_normal() * 0.5
        res["gin_out"] = np.asarray(gin(x, adj=adjs)); res["gin_eps"] = np.asarray([float(e) for e in gin.epsilon])
        # --- GraphGather (sums ALL rows, padding included: quirk Q4), GraphMaxPooling, GAT
        res["gather_out"] = np.asarray(layers.GraphGather()(tf_standin.t(res["conv_out"])))
        res["maxpool_out"] = np.asarray(layers.GraphMaxPooling(C)(x, adj=adjs))
        gat = layers.GAT(C)
        res["gat_out"] = np.asarray(gat(x, adj=adjs)); res["gat_a"] = np.stack([np.asarray(a) for a in gat.weight_a])
        # --- GraphDense: dense path, and the ragged path with per-graph true sizes
        dn = layers.GraphDense(6)
        dn(x)
        dn.bias[...] = rng.standard_normal(dn.bias.shape) * 0.2
        res["dense_out"] = np.asarray(dn(x)); res["dense_k"] = np.asarray(dn.kernel); res["dense_b"] = np.asarray(dn.bias)
        sizes = np.asarray([10, 7, 1, 0, 10, 3, 9, 10, 2, 5] + [0] * 20, np.int32)
        xr = tf_standin.t(rng.standard_normal(x.shape))       # values in the rows BEYOND the true size must not matter
        dr = layers.GraphDense(6)
        dr(xr, enabled_node_nums=sizes, max_node_num=10)
        dr.kernel[...] = np.asarray(dn.kernel); dr.bias[...] = np.asarray(dn.bias)
        res["ragged_x"] = np.asarray(xr); res["ragged_sizes"] = sizes
        res["ragged_dense_out"] = np.asarray(dr(xr, enabled_node_nums=sizes, max_node_num=10))
        # --- GraphBatchNormalization as the model files call it (learning phase 0: moving statistics), both paths
        bn = layers.GraphBatchNormalization()
        res["bn_out"] = np.asarray(bn(xr, max_node_num=10))
        bnr = layers.GraphBatchNormalization()
        res["bn_ragged_out"] = np.asarray(bnr(xr, enabled_node_nums=sizes, max_node_num=10))
        out[tag] = res
        np.savez_compressed(os.path.join(HERE, "g6_layers_%s.npz" % tag), **res)
        print(tag, {k: v.shape for k, v in res.items() if hasattr(v, "shape") and k.endswith("_out")})


if __name__ == "__main__":
    main()
