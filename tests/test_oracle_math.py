"""Self-checks of the oracle's arithmetic half (no reference outputs exist for it -- "parity
unpinned" at TF): the four GraphConv dispatch branches agree, gradients match fp64 finite
differences, fast block-diagonal forms equal the per-graph loops, SpMM edge semantics."""
import numpy as np
import pytest

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_oracle as K


def _batch(channels="plain"):
    z = load_golden("g3_synthetic_feed_b30.npz")
    adjs = unflatten_adjs(z, "adj_")
    x = z["features"].astype(np.float64)
    if channels == "split":
        raw = load_golden("g1_synthetic_raw.npz")
        full, _, _ = K.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64),
                                   "max_node_num": 10}, split_adj_flag=True)
        fb = K.feed_batch(list(z["batch_idx"]), 30, full)
        adjs = fb["adjs"]
    return x, adjs


@pytest.mark.parametrize("channels", ["plain", "split"])
def test_four_variants_agree(channels):
    x, adjs = _batch(channels)
    rng = np.random.default_rng(0)
    C = len(adjs[0])
    w, b = K.graphconv_params(rng, 3, 50, C)
    b = [rng.standard_normal((1, 50)) for _ in range(C)]
    ref = K.graphconv_fwd(x, adjs, w, b, "default")
    for v in ("bspmm", "bconv", "batched"):
        assert np.abs(K.graphconv_fwd(x, adjs, w, b, v) - ref).max() < 1e-12, v
    assert np.abs(K.graphconv_fwd_fast(x, adjs, w, b) - ref).max() < 1e-12
    # padded dummy graphs (b >= 10) have empty adjacency -> all-zero output rows
    assert np.all(ref[10:] == 0)


def test_graphconv_grad_finite_difference():
    x, adjs = _batch("split")
    x = x[:12]
    adjs = adjs[:12]
    rng = np.random.default_rng(1)
    C = len(adjs[0])
    w = [rng.standard_normal((3, 7)) for _ in range(C)]
    b = [rng.standard_normal((1, 7)) for _ in range(C)]
    g = rng.standard_normal((12, 10, 7))
    x = x + 0.1 * rng.standard_normal(x.shape)

    def loss(x_, w_, b_, adjs_=adjs):
        return float((K.graphconv_fwd(x_, adjs_, w_, b_) * g).sum())

    dx, dw, db, dv = K.graphconv_bwd(x, adjs, w, b, g, want_values_grad=True)
    dx2, dw2, db2 = K.graphconv_bwd_fast(x, adjs, w, g)
    assert np.abs(dx - dx2).max() < 1e-12
    for c in range(C):
        assert np.abs(dw[c] - dw2[c]).max() < 1e-12 and np.abs(db[c] - db2[c]).max() < 1e-12
    h = 1e-6
    for _ in range(20):
        i = tuple(rng.integers(0, s) for s in x.shape)
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        assert abs((loss(xp, w, b) - loss(xm, w, b)) / (2 * h) - dx[i]) < 1e-6
    for c in range(C):
        for _ in range(5):
            i = tuple(rng.integers(0, s) for s in w[c].shape)
            wp = [t.copy() for t in w]
            wm = [t.copy() for t in w]
            wp[c][i] += h
            wm[c][i] -= h
            assert abs((loss(x, wp, b) - loss(x, wm, b)) / (2 * h) - dw[c][i]) < 1e-6
            j = int(rng.integers(0, 7))
            bp = [t.copy() for t in b]
            bm = [t.copy() for t in b]
            bp[c][0, j] += h
            bm[c][0, j] -= h
            assert abs((loss(x, w, bp) - loss(x, w, bm)) / (2 * h) - db[c][0, j]) < 1e-6
    # gradient w.r.t. adjacency values (bspmm_call.py:50-55)
    bsel, csel = 3, 5
    idx, val, shape = adjs[bsel][csel]
    for e in range(len(val)):
        def with_val(d):
            a2 = [list(r) for r in adjs]
            v2 = np.asarray(val, np.float64).copy()
            v2[e] += d
            a2[bsel][csel] = (idx, v2, shape)
            return a2
        fd = (loss(x, w, b, with_val(h)) - loss(x, w, b, with_val(-h))) / (2 * h)
        assert abs(fd - dv[bsel][csel][e]) < 1e-6


def test_op_contracts_and_grads():
    x, adjs = _batch("plain")
    rng = np.random.default_rng(2)
    T = 10
    al = [adjs[t][0] for t in range(T)]
    rhs = [rng.standard_normal((10, 5)) for _ in range(T)]
    g = [rng.standard_normal((10, 5)) for _ in range(T)]
    out = K.bspmm(al, rhs)
    dense = [np.zeros((10, 10)) for _ in range(T)]
    for t in range(T):
        np.add.at(dense[t], (al[t][0][:, 0], al[t][0][:, 1]), al[t][1])
        assert np.abs(out[t] - dense[t] @ rhs[t]).max() < 1e-12
    vg, rg = K.bspmm_grad(al, rhs, g)
    for t in range(T):
        assert np.abs(rg[t] - dense[t].T @ g[t]).max() < 1e-12
        full = g[t] @ rhs[t].T
        assert np.abs(vg[t] - full[al[t][0][:, 0], al[t][0][:, 1]]).max() < 1e-12
    # adjoint forms
    o2 = K.bspmm(al, [r.T.copy() for r in rhs], adjoint_a=True, adjoint_b=True)
    for t in range(T):
        assert np.abs(o2[t] - dense[t].T @ rhs[t]).max() < 1e-12
    # bspmdt == bspmm on row slices; its rhs gradient is the stacked one
    o3 = K.bspmdt(al, np.concatenate(rhs, 0))
    assert np.abs(np.stack(o3) - np.stack(out)).max() == 0
    _, rg3 = K.bspmdt_grad(al, np.concatenate(rhs, 0), g)
    assert np.abs(rg3 - np.concatenate(rg, 0)).max() == 0
    # bconv: channel add-n
    o4 = K.bconv([[a, a] for a in al], [[r, 2 * r] for r in rhs])
    assert np.abs(np.stack(o4) - 3 * np.stack(out)).max() < 1e-12


def test_spmm_duplicates_unsorted_and_empty():
    idx = np.array([[2, 1], [0, 0], [2, 1], [1, 2]], np.int32)     # unsorted + duplicate
    val = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    rhs = np.arange(12, dtype=np.float64).reshape(3, 4)
    out = K.spmm_coo((idx, val, [4, 3]), rhs)
    dense = np.zeros((4, 3))
    dense[2, 1] = 4
    dense[0, 0] = 2
    dense[1, 2] = 4
    assert np.array_equal(out, dense @ rhs)
    assert np.all(out[3] == 0)                                      # row without entries
    e = K.spmm_coo((np.zeros((0, 2), np.int32), np.zeros((0,), np.float32), [4, 3]), rhs)
    assert e.shape == (4, 4) and np.all(e == 0)


def test_dense_gin_gather():
    rng = np.random.default_rng(3)
    x, adjs = _batch("plain")
    x = x + rng.standard_normal(x.shape)
    k = rng.standard_normal((3, 6))
    bias = rng.standard_normal(6)
    y = K.graphdense_fwd(x, k, bias)
    assert y.shape == (30, 10, 6)
    g = rng.standard_normal(y.shape)
    dx, dk, dbias = K.graphdense_bwd(x, k, g)
    h = 1e-6
    kp = k.copy(); kp[1, 2] += h
    km = k.copy(); km[1, 2] -= h
    fd = ((K.graphdense_fwd(x, kp, bias) - K.graphdense_fwd(x, km, bias)) * g).sum() / (2 * h)
    assert abs(fd - dk[1, 2]) < 1e-6
    assert np.allclose(dbias, g.reshape(-1, 6).sum(0))
    en = np.array([10] * 10 + [0] * 20)
    yr = K.graphdense_ragged_fwd(x, k, bias, en)
    assert np.all(yr[10:] == 0) and np.array_equal(yr[:10], y[:10])
    # GIN: default branch keeps eps, accelerated branches drop it (quirk Q1)
    eps = [0.3]
    o = K.gin_fwd(x, adjs, eps)
    o0 = K.gin_fwd(x, adjs, eps, with_eps=False)
    assert np.allclose(o - o0, 0.3 * x)
    gg = rng.standard_normal(x.shape)
    dxg, deps = K.gin_bwd(x, adjs, eps, gg)
    fd = ((K.gin_fwd(x, adjs, [0.3 + h]) - K.gin_fwd(x, adjs, [0.3 - h])) * gg).sum() / (2 * h)
    assert abs(fd - deps[0]) < 1e-5
    xp = x.copy(); xp[2, 3, 1] += h
    xm = x.copy(); xm[2, 3, 1] -= h
    fd = ((K.gin_fwd(xp, adjs, eps) - K.gin_fwd(xm, adjs, eps)) * gg).sum() / (2 * h)
    assert abs(fd - dxg[2, 3, 1]) < 1e-6
    s = K.gather_fwd(x)
    assert np.allclose(s, x.sum(1)) and K.gather_bwd(s, 10).shape == x.shape


def test_synth_generator_cfg2():
    adjs = K.synth_mol_graphs(np.random.default_rng(1234), 50, 32, 3)
    for a in adjs:
        idx, val, shape = a[0]
        assert len(val) == 100 and list(shape) == [32, 32]
        d = np.zeros((32, 32))
        d[idx[:, 0], idx[:, 1]] = val
        assert np.array_equal(d, d.T) and np.all(np.diag(d) == 1)
        assert np.all(np.diff(idx[:, 0] * 32 + idx[:, 1]) > 0)      # row-major sorted


def test_graph_maxpool_matches_densified_definition_and_fd():
    """kgcn/layers.py:122-150 restated literally (densify A .* x_k, reduce_max over columns) vs the
    sparse form of the oracle; gradient against central finite differences."""
    rng = np.random.default_rng(5)
    B, N, D = 3, 6, 4
    adjs = []
    for b in range(B):
        chans = []
        for c in range(2):
            dense = (rng.random((N, N)) < 0.4) * rng.standard_normal((N, N))
            if b == 0 and c == 0:
                dense[2] = rng.standard_normal(N)            # a FULL row: no implicit zero candidate
            idx = np.argwhere(dense != 0).astype(np.int32)
            chans.append((idx, dense[dense != 0].astype(np.float32), [N, N]))
        adjs.append(chans)
    x = rng.standard_normal((B, N, D))
    out = K.graph_maxpool_fwd(x, adjs)
    ref = np.zeros_like(out)
    for b in range(B):
        for c in range(2):
            idx, val, _ = adjs[b][c]
            dense = np.zeros((N, N))
            dense[idx[:, 0], idx[:, 1]] = val
            for k in range(D):
                ref[b, :, k] += (dense * x[b, :, k][None, :]).max(axis=1)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12)
    g = rng.standard_normal(out.shape)
    dx = K.graph_maxpool_bwd(x, adjs, g)
    eps = 1e-6
    for _ in range(10):
        i = tuple(rng.integers(0, s) for s in x.shape)
        xp, xm = x.copy(), x.copy()
        xp[i] += eps
        xm[i] -= eps
        fd = ((K.graph_maxpool_fwd(xp, adjs) - K.graph_maxpool_fwd(xm, adjs)) * g).sum() / (2 * eps)
        assert abs(fd - dx[i]) < 1e-6


def test_graph_maxpool_tie_split():
    """tf.reduce_max's gradient divides equally among all maximal elements of the densified row,
    the implicit zeros included."""
    N = 4
    adj = [[(np.array([[0, 1], [0, 2], [1, 0]], np.int32), np.array([1.0, 1.0, 1.0], np.float32), [N, N])]]
    x = np.array([[[5.0], [2.0], [2.0], [7.0]]])            # row 0: two entries tie at 2 > 0
    g = np.ones((1, N, 1))
    dx = K.graph_maxpool_bwd(x, adj, g)
    np.testing.assert_allclose(dx[0, :, 0], [1.0, 0.5, 0.5, 0.0])
    x0 = np.array([[[-1.0], [0.0], [-3.0], [7.0]]])         # row 0: entry value 0 ties with 2 implicit zeros
    dx0 = K.graph_maxpool_bwd(x0, adj, g)
    np.testing.assert_allclose(dx0[0, :, 0], [0.0, 1.0 / 3.0, 0.0, 0.0])


def test_integrated_gradients_completeness():
    """Oracle of the attribution loop (kgcn/visualization.py:187-275): with features and adjacency values
    both scaled, the sum of the integrated gradients approaches score(1) - score(0)."""
    rng = np.random.default_rng(9)
    B, N, F, Dh = 2, 6, 3, 4
    adjs = K.normalize_adj(K.synth_mol_graphs(rng, B, N, 1))
    x = rng.standard_normal((B, N, F))
    w, b = [rng.standard_normal((F, Dh)) * 0.5], [rng.standard_normal((1, Dh)) * 0.1]
    ro = rng.standard_normal(Dh)
    ig_x, ig_a = K.integrated_gradients(x, adjs, w, b, ro, 400)
    zero_adjs = [[(m[0], np.zeros_like(m[1]), m[2]) for m in chs] for chs in adjs]
    s1 = K.probe_score(x, adjs, w, b, ro)[0]
    s0 = K.probe_score(x * 0, zero_adjs, w, b, ro)[0]
    total = ig_x.sum() + sum(a.sum() for a in ig_a)
    assert abs(total - (s1 - s0)) < 5e-3 * max(1.0, abs(s1 - s0)), (total, s1 - s0)


def test_gat_literal_and_finite_differences():
    """GAT oracle vs the reference's literal formulation (one-hot matmuls, kgcn/layers.py:517-533) and
    its gradients vs central differences."""
    rng = np.random.default_rng(4)
    B, N, D, C = 2, 7, 3, 2
    adjs = []
    for b in range(B):
        chans = []
        for c in range(C):
            dense = (rng.random((N, N)) < 0.35).astype(np.float32)
            dense[5] = 0                                            # an empty row: sigmoid(0) = 0.5
            idx = np.argwhere(dense != 0).astype(np.int32)
            chans.append((idx, np.ones(len(idx), np.float32), [N, N]))
        adjs.append(chans)
    x = rng.standard_normal((B, N, D))
    wa = [rng.standard_normal((2 * D, 1)) * 0.7 for _ in range(C)]
    out = K.gat_fwd(x, adjs, wa)
    ref = np.zeros_like(out)
    for b in range(B):
        for c in range(C):
            idx = adjs[b][c][0]
            a1, a2 = x[b][idx[:, 1]], x[b][idx[:, 0]]
            ii = np.eye(N)[idx[:, 0]].T                             # tf.transpose(tf.one_hot(idx[:,0], N))
            layer = np.concatenate([a1, a2], axis=1) @ wa[c]
            layer = np.where(layer > 0, layer, 0.2 * layer)
            e = np.exp(layer)
            denom = ii @ e
            alpha = e / (denom[idx[:, 1]] + 1.0e-10)
            ref[b] += 1.0 / (1.0 + np.exp(-(ii @ (alpha * a1))))
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-12)
    assert np.all(out[:, 5] == 0.5 * C)
    g = rng.standard_normal(out.shape)
    dx, dwa = K.gat_bwd(x, adjs, wa, g)
    h = 1e-6
    for _ in range(8):
        i = tuple(rng.integers(0, s) for s in x.shape)
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        fd = ((K.gat_fwd(xp, adjs, wa) - K.gat_fwd(xm, adjs, wa)) * g).sum() / (2 * h)
        assert abs(fd - dx[i]) < 1e-6 * max(1.0, abs(fd)), (i, fd, dx[i])
    for c in range(C):
        for k in (0, D - 1, D, 2 * D - 1):
            wp = [w.copy() for w in wa]
            wm = [w.copy() for w in wa]
            wp[c][k, 0] += h
            wm[c][k, 0] -= h
            fd = ((K.gat_fwd(x, adjs, wp) - K.gat_fwd(x, adjs, wm)) * g).sum() / (2 * h)
            assert abs(fd - dwa[c][k, 0]) < 1e-6 * max(1.0, abs(fd)), (c, k, fd, dwa[c][k, 0])


def test_gram_decoder_gradients():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 5, 4))
    w = rng.standard_normal(4)
    g = rng.standard_normal((3, 5, 5))
    out = K.gram_fwd(x, w)
    np.testing.assert_allclose(out[1], (x[1] * w) @ x[1].T, atol=1e-12)
    dx, dw = K.gram_bwd(x, w, g)
    h = 1e-6
    for i in [(0, 1, 2), (2, 4, 0)]:
        xp, xm = x.copy(), x.copy()
        xp[i] += h; xm[i] -= h
        fd = ((K.gram_fwd(xp, w) - K.gram_fwd(xm, w)) * g).sum() / (2 * h)
        assert abs(fd - dx[i]) < 1e-6 * max(1, abs(fd))
    for k in range(4):
        wp, wm = w.copy(), w.copy()
        wp[k] += h; wm[k] -= h
        fd = ((K.gram_fwd(x, wp) - K.gram_fwd(x, wm)) * g).sum() / (2 * h)
        assert abs(fd - dw[k]) < 1e-6 * max(1, abs(fd))
    dx0, none = K.gram_bwd(x, None, g)
    assert none is None and np.allclose(dx0, K.gram_bwd(x, np.ones(4), g)[0])
