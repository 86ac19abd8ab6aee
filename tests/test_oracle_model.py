"""Self-check of the model-level oracle (oracle/kgcn_model_oracle.py): hand-written backward vs
finite differences, and that a few Adam steps on synthetic.jbl reduce the loss."""
import numpy as np
import pytest

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_model_oracle as M


def _batch():
    z = load_golden("g3_synthetic_feed_full30.npz")
    adjs = unflatten_adjs(z, "adj_")
    return z["features"].astype(np.float64), adjs, z["labels"].astype(np.float64), z["mask"].astype(np.float64)


def test_model_backward_finite_difference():
    x, adjs, labels, mask = _batch()
    rng = np.random.default_rng(0)
    p = M.init_params(rng, 3)
    for k in ("b1", "b2", "b3"):
        p[k] = [rng.standard_normal((1, 50)) * 0.1]
    c = M.forward(p, x, adjs, labels, mask)
    g = M.backward(p, c, x, adjs, labels, mask)
    h = 1e-6
    for key, idx in [("w1", (1, 7)), ("w2", (3, 4)), ("w3", (10, 2)), ("b2", (0, 5)), ("dk", (4, 4)),
                     ("db", (3,)), ("ok", (2, 1)), ("ob", (0,)), ("gamma", (6,)), ("beta", (9,))]:
        def at(d):
            q = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in p.items()}
            t = q[key][0] if isinstance(q[key], list) else q[key]
            t[idx] += d
            return M.forward(q, x, adjs, labels, mask)["cost_opt"]
        fd = (at(h) - at(-h)) / (2 * h)
        an = (g[key][0] if isinstance(g[key], list) else g[key])
        an = np.asarray(an).reshape((p[key][0] if isinstance(p[key], list) else p[key]).shape)[idx]
        assert abs(fd - an) < 1e-7 + 1e-4 * abs(an), (key, fd, an)


def test_adam_steps_reduce_loss():
    x, adjs, labels, mask = _batch()
    p = M.init_params(np.random.default_rng(1), 3)
    opt = M.TFAdam(lr=0.01)
    losses = []
    for _ in range(30):
        p, c = M.train_step(p, opt, x, adjs, labels, mask)
        losses.append(c["cost_opt"])
    assert losses[-1] < losses[0] - 0.01, losses[::5]


# ---- the two other named model files (oracle/kgcn_nets_oracle.py) -----------------------------------
from oracle import kgcn_nets_oracle as NETS
from oracle import kgcn_oracle as K


def _fd_check(p, loss_at, grads, probes, h=1e-6):
    for key, idx in probes:
        def at(d):
            q = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in p.items()}
            t = q[key][0] if isinstance(q[key], list) else q[key]
            t[idx] += d
            return loss_at(q)
        fd = (at(h) - at(-h)) / (2 * h)
        ref = p[key][0] if isinstance(p[key], list) else p[key]
        an = np.asarray(grads[key][0] if isinstance(grads[key], list) else grads[key]).reshape(ref.shape)[idx]
        assert abs(fd - an) < 1e-7 + 1e-4 * abs(an), (key, fd, an)


def tox21_like_batch(rng, B=6, N=12, F=7, T=3):
    sizes = rng.integers(3, N + 1, size=B)
    sizes[-1] = 0                                            # a dummy (padding) graph, mask 0
    adjs, x = [], np.zeros((B, N, F))
    for b, n in enumerate(sizes):
        if n == 0:
            adjs.append([(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [N, N])])
            continue
        a = K.synth_mol_graphs(rng, 1, int(n), 1)[0][0]
        adjs.append([(a[0], a[1], [N, N])])
        x[b, :n] = rng.standard_normal((n, F))
    labels = rng.integers(0, 2, size=(B, T)).astype(np.float64)
    mask_label = (rng.random((B, T)) < 0.8).astype(np.float64)
    mask = (sizes > 0).astype(np.float64)
    return x, K.normalize_adj(adjs), labels, mask, mask_label, sizes


@pytest.mark.parametrize("pos_weight", [None, 3.0, "per_task"])
def test_multitask_backward_finite_difference(pos_weight):
    rng = np.random.default_rng(11)
    x, adjs, labels, mask, mask_label, sizes = tox21_like_batch(rng)
    p = NETS.multitask_init(rng, x.shape[2], labels.shape[1], widths=(9, 8, 7, 6, 5))
    for k in ("b1", "b2", "b4"):
        p[k] = [rng.standard_normal(p[k][0].shape) * 0.1]
    p["beta"] = rng.standard_normal(p["beta"].shape) * 0.1
    if isinstance(pos_weight, str):          # info.pos_weight of the reference: one weight per label column (kgcn/data_util.py:563-568)
        pos_weight = rng.uniform(0.5, 4.0, size=labels.shape[1])
    args = (x, adjs, labels, mask, mask_label, sizes, pos_weight)
    c = NETS.multitask_forward(p, *args)
    g = NETS.multitask_backward(p, c, x, adjs, labels, mask, mask_label, pos_weight)
    assert np.all(c["bn"][-1] == 0) and np.all(c["bn"][0, sizes[0]:] == 0)        # padding rows are zero
    _fd_check(p, lambda q: NETS.multitask_forward(q, *args)["cost_opt"], g,
              [("w1", (1, 2)), ("b1", (0, 3)), ("w2", (4, 4)), ("k3", (2, 5)), ("c3", (1,)), ("w4", (3, 3)),
               ("b4", (0, 2)), ("gamma", (4,)), ("beta", (0,)), ("k5", (5, 1)), ("ok", (2, 1)), ("ob", (0,))])


def _sparse_batch(rng, nmol=5, F=6, max_degree=0, normalize=True):
    ex, sizes = [], []
    for _ in range(nmol):
        n = int(rng.integers(3, 9))
        idx, val, _ = K.synth_mol_graphs(rng, 1, n, 1)[0][0]
        a = np.zeros((n, n), np.float32)
        a[idx[:, 0], idx[:, 1]] = val
        feat = (rng.random((n, F)) < 0.5) * rng.standard_normal((n, F))
        ex.append(K.sparse_example(a, feat.astype(np.float32)))
        sizes.append(n)
    f = K.collate_sparse_examples(ex)
    return f, np.array(sizes)


def _construct(f, F, **kw):
    return K.construct_batched_adjacency_and_feature_matrices(
        f["size"][:, 0], f["adj_row"], f["adj_column"], f["adj_values"], f["adj_elem_len"], f["adj_degrees"],
        f["feature_row"], f["feature_column"], f["feature_values"], f["feature_elem_len"], F, **kw)


@pytest.mark.parametrize("mode", ["normalize", "split"])
def test_sparse_model_backward_finite_difference(mode):
    rng = np.random.default_rng(21)
    F = 6
    f, sizes = _sparse_batch(rng, F=F)
    kw = dict(max_degree=0, normalize=True) if mode == "normalize" else dict(max_degree=5, normalize=False, split_adj=True)
    chans, net = _construct(f, F, **kw)
    assert len(chans) == (1 if mode == "normalize" else 6)
    labels = rng.integers(0, 3, size=len(sizes))
    p = NETS.sparse_init(rng, F, 3, channels=len(chans), out_dims=(7, 6), dense_dim=5)
    p["b1"] = [rng.standard_normal(b.shape) * 0.1 for b in p["b1"]]
    c = NETS.sparse_forward(p, net, chans, sizes, labels)
    g = NETS.sparse_backward(p, c, chans, sizes, labels)
    _fd_check(p, lambda q: NETS.sparse_forward(q, net, chans, sizes, labels)["loss"], g,
              [("w1", (1, 2)), ("b1", (0, 3)), ("w2", (4, 4)), ("dk", (2, 3)), ("dc", (1,)), ("gamma", (4,)),
               ("beta", (0,)), ("ok", (2, 1)), ("ob", (0,))])


def test_block_diagonal_known_answer_from_reference_docstring():
    """The worked example in the reference's own docstring (kgcn/data_util.py:703-733): two molecules of
    2 and 3 nodes -> the [5,5] block-diagonal matrix and the [5,10] feature matrix printed there."""
    chans, net = K.construct_batched_adjacency_and_feature_matrices(
        [2, 3], [0, 0, 1, 1, 0, 0, 1, 1, 1, 2, 2], [0, 1, 0, 1, 0, 1, 0, 1, 2, 1, 2], np.ones(11, np.float32), [4, 7],
        np.zeros(11, np.int64), [0, 1, 0, 1, 2], [2, 3, 1, 2, 3], [4, 5, 1, 2, 3], [2, 3], 10, normalize=False,
        split_adj=False)
    idx, val, shape = chans[0]
    dense = np.zeros(shape)
    dense[idx[:, 0], idx[:, 1]] = val
    np.testing.assert_array_equal(dense, [[1, 1, 0, 0, 0], [1, 1, 0, 0, 0], [0, 0, 1, 1, 0], [0, 0, 1, 1, 1], [0, 0, 0, 1, 1]])
    expect = np.zeros((5, 10))
    expect[0, 2], expect[1, 3], expect[2, 1], expect[3, 2], expect[4, 3] = 4, 5, 1, 2, 3
    np.testing.assert_array_equal(net, expect)


def test_gin_model_backward_finite_difference():
    """oracle/kgcn_nets_oracle.gin_* (example_model/model_gin.py:40-78; the checker of the full-size cfg5 GPU test) against
    central differences of its own forward, and its aggregation against the per-graph K.gin_fwd."""
    rng = np.random.default_rng(3)
    B, N, F, W = 7, 10, 6, 5
    adjs = K.synth_ring_graphs(rng, B, N)
    x = rng.standard_normal((B, N, F))
    labels = np.eye(2)[rng.integers(0, 2, B)]
    mask = (rng.random(B) < 0.8).astype(np.float64)
    p = NETS.gin_init(rng, F, W)
    c = NETS.gin_forward(p, x, adjs, labels, mask)
    np.testing.assert_allclose(c["a0"].reshape(B, N, F), K.gin_fwd(x, adjs, p["eps"][0]), rtol=0, atol=1e-12)
    g = NETS.gin_backward(p, c, x, adjs, labels, mask)
    f = lambda q: NETS.gin_forward(q, x, adjs, labels, mask)["cost_opt"]
    h = 1e-6
    for k, idx in [("k0", (1, 2)), ("c0", (3,)), ("k1", (0, 4)), ("k2", (2, 2)), ("c3", (1,)), ("ok", (7, 1)), ("ob", (0,))]:
        q = {kk: (v.copy() if hasattr(v, "copy") else [e.copy() for e in v]) for kk, v in p.items()}
        q[k][idx] += h; up = f(q); q[k][idx] -= 2 * h; dn = f(q)
        assert abs((up - dn) / (2 * h) - g[k][idx]) < 1e-6 * max(1.0, abs(g[k][idx])), (k, idx)
    for blk in range(2):
        q = {kk: (v.copy() if hasattr(v, "copy") else [e.copy() for e in v]) for kk, v in p.items()}
        q["eps"][blk][0] += h; up = f(q); q["eps"][blk][0] -= 2 * h; dn = f(q)
        assert abs((up - dn) / (2 * h) - g["eps"][blk][0]) < 1e-6 * max(1.0, abs(g["eps"][blk][0]))
    xq = x.copy(); xq[2, 3, 1] += h; up = NETS.gin_forward(p, xq, adjs, labels, mask)["cost_opt"]
    xq[2, 3, 1] -= 2 * h; dn = NETS.gin_forward(p, xq, adjs, labels, mask)["cost_opt"]
    assert abs((up - dn) / (2 * h) - g["dx"][2, 3, 1]) < 1e-6
