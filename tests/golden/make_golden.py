#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's numpy half.

Runs only in the build container (needs /root/reference).  Nothing of the reference is copied:
only the arrays its functions return are saved (small .npz files).  The reference's arithmetic
half (TensorFlow) cannot be imported here, so these fixtures pin the *harness* side of the hot
path (what feeds GraphConv): SURVEY.md section 8c, G1..G4.

The reference modules import tensorflow at module top; a 3-attribute stub module satisfies
kgcn/data_util.py:5-8 and kgcn/feed.py:1-4,122,126 (SparseTensorValue is just a namedtuple there).

    python tests/golden/make_golden.py
"""
import collections
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "1.15.0"
    tf.SparseTensorValue = collections.namedtuple(
        "SparseTensorValue", ["indices", "values", "dense_shape"])
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF)
    from kgcn import data_util, feed  # noqa: E402  (reference code, imported in place)
    return data_util, feed


def _flatten_adjs(adjs):
    """adjs[g][ch] = (idx [nnz,2], val [nnz], shape[2]) -> flat arrays with offsets."""
    G, C = len(adjs), len(adjs[0])
    idx, val, shp, off = [], [], [], [0]
    for g in range(G):
        assert len(adjs[g]) == C
        for ch in range(C):
            i, v, s = adjs[g][ch]
            i = np.asarray(i).reshape(-1, 2)
            idx.append(i.astype(np.int64))
            val.append(np.asarray(v))
            shp.append(np.asarray(s, dtype=np.int64))
            off.append(off[-1] + i.shape[0])
    vals = np.concatenate(val)
    return dict(idx=np.concatenate(idx).astype(np.int32), val=vals,
                val_dtype=str(vals.dtype),
                idx_dtype=str(np.asarray(adjs[0][0][0]).dtype),
                shape=np.stack(shp), offsets=np.asarray(off, np.int64),
                num_graphs=G, num_channels=C)


def main():
    data_util, feed = _import_reference()
    import joblib

    def load(cfg, name):
        with contextlib.redirect_stdout(io.StringIO()):
            return data_util.load_data(cfg, os.path.join(REF, "example_jbl", name),
                                       prohibit_shuffle=True)

    base = {"with_feature": True, "with_node_embedding": False, "normalize_adj_flag": False,
            "split_adj_flag": False, "order": 1, "shuffle_data": False,
            "task": "multitask_classification"}

    # ---- G1: the raw dataset (inputs), as stored in the reference's own data file ----------
    raw = joblib.load(os.path.join(REF, "example_jbl", "synthetic.jbl"))
    np.savez_compressed(
        os.path.join(HERE, "g1_synthetic_raw.npz"),
        feature=raw["feature"].astype(np.float64),
        dense_adj=raw["dense_adj"].astype(np.int8),
        label=raw["label"].astype(np.float64),
        mask_label=raw["mask_label"].astype(np.int64),
        max_node_num=np.int64(raw["max_node_num"]))

    # ---- G2: reference build_data adjacency for the flag combinations ---------------------
    for tag, upd in [("plain", {}), ("norm", {"normalize_adj_flag": True}),
                     ("split", {"split_adj_flag": True}),
                     ("split_norm", {"split_adj_flag": True, "normalize_adj_flag": True}),
                     ("order2", {"order": 2})]:
        cfg = dict(base, **upd)
        all_data, info = load(cfg, "synthetic.jbl")
        out = _flatten_adjs(all_data.adjs)
        out["adj_channel_num"] = np.int64(info.adj_channel_num)
        out["enabled_node_nums"] = np.asarray(all_data.enabled_node_nums)
        np.savez_compressed(os.path.join(HERE, "g2_synthetic_adj_%s.npz" % tag), **out)

    # ---- G3: one padded batch exactly as construct_feed emits it (10 real + 20 dummy) ------
    all_data, info = load(base, "synthetic.jbl")
    B = 30
    placeholders = {"adjs": [[("adj", ch, b) for ch in range(info.adj_channel_num)]
                             for b in range(B)],
                    "features": "features", "labels": "labels", "mask": "mask",
                    "enabled_node_nums": "enabled_node_nums"}
    batch_idx = list(range(150, 160))
    fd = feed.construct_feed(batch_idx, placeholders, all_data, batch_size=B, info=info,
                             config=base)
    fadj = [[fd[("adj", ch, b)] for ch in range(info.adj_channel_num)] for b in range(B)]
    # dummy entries carry b_shape of the last real graph (feed.py:116-126)
    flat = _flatten_adjs([[(a.indices, a.values, a.dense_shape) for a in row] for row in fadj])
    np.savez_compressed(
        os.path.join(HERE, "g3_synthetic_feed_b30.npz"),
        batch_idx=np.asarray(batch_idx, np.int64), batch_size=np.int64(B),
        features=fd["features"], labels=fd["labels"], mask=fd["mask"],
        enabled_node_nums=fd["enabled_node_nums"],
        **{"adj_" + k: v for k, v in flat.items()})
    # a full batch too (30 real graphs, the shape cfg1 runs 5 of 6 iterations with)
    batch_idx = list(range(0, 30))
    fd = feed.construct_feed(batch_idx, placeholders, all_data, batch_size=B, info=info,
                             config=base)
    fadj = [[fd[("adj", ch, b)] for ch in range(info.adj_channel_num)] for b in range(B)]
    flat = _flatten_adjs([[(a.indices, a.values, a.dense_shape) for a in row] for row in fadj])
    np.savez_compressed(
        os.path.join(HERE, "g3_synthetic_feed_full30.npz"),
        batch_idx=np.asarray(batch_idx, np.int64), batch_size=np.int64(B),
        features=fd["features"], labels=fd["labels"], mask=fd["mask"],
        enabled_node_nums=fd["enabled_node_nums"],
        **{"adj_" + k: v for k, v in flat.items()})

    # ---- G4: the sparse-COO + node-id form (synthetic_sparse.jbl) -------------------------
    # The shipped file lacks the "node_num" key build_data reads in node-embedding mode
    # (data_util.py:496), so the dict is completed with it (an input field, max id + 1)
    # before calling the reference's build_data.
    cfg = dict(base, with_feature=False, with_node_embedding=True)
    raws = joblib.load(os.path.join(REF, "example_jbl", "synthetic_sparse.jbl"))
    raws = dict(raws)
    raws["node_num"] = int(np.max(np.asarray(raws["node"]))) + 1
    with contextlib.redirect_stdout(io.StringIO()):
        all_data, info = data_util.build_data(cfg, raws, prohibit_shuffle=True)
    out = _flatten_adjs(all_data.adjs)
    out["nodes"] = np.asarray(all_data.nodes)
    out["enabled_node_nums"] = np.asarray(all_data.enabled_node_nums)
    # raw inputs, flattened the same way, so the build's loader can be run on them
    rin = _flatten_adjs([[a] for a in raws["adj"]])
    out.update({"in_" + k: v for k, v in rin.items()})
    out["in_max_node_num"] = np.int64(raws["max_node_num"])
    np.savez_compressed(os.path.join(HERE, "g4_synthetic_sparse_loader.npz"), **out)

    # ---- multi-channel raw input (multi_dense_adj, C=2) and its reference COO --------------
    rawm = joblib.load(os.path.join(REF, "example_jbl", "sample_multiadj.jbl"))
    all_data, info = load(base, "sample_multiadj.jbl")
    out = _flatten_adjs(all_data.adjs)
    out["multi_dense_adj"] = np.asarray(rawm["multi_dense_adj"], dtype=np.float64)
    out["feature"] = np.asarray(rawm["feature"], dtype=np.float64)
    out["enabled_node_nums"] = np.asarray(all_data.enabled_node_nums)
    np.savez_compressed(os.path.join(HERE, "g2_sample_multiadj.npz"), **out)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-40s %7d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
