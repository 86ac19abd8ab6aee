"""oracle/kgcn_ref.c (the C restatement that checks every FULL-SIZE GPU test and is timed as bench.py's cpu_baseline) against
oracle/kgcn_oracle.py (the numpy restatement the small tests and the golden fixtures use) -- off the GPU, so the full-size
checker is itself checked: GraphConv forward / backward, Bspmm (+ adjoint), dense forward / backward for all four activation
codes, GINAggregate (+ adjoint, + the d epsilon dot product).  fp32 C against fp64 numpy: tolerance = fp32 rounding at these sizes."""
import numpy as np
import pytest

from oracle import kgcn_oracle as K
from oracle import ref_c


def _batch(rng, T, n, normalize):
    adjs = K.synth_mol_graphs(rng, T, n, 3, normalize=normalize)
    adjs[T // 2] = [(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [n, n])]        # a dummy graph (kgcn/feed.py:123-126)
    return adjs, ref_c.flatten_coo([a[0] for a in adjs])


def _close(got, ref, rel, what):
    ref = np.asarray(ref, np.float64)
    err = float(np.abs(np.asarray(got, np.float64) - ref).max())
    assert err <= rel * max(1e-30, float(np.abs(ref).max())), "%s: %.3e of %.3e" % (what, err, np.abs(ref).max())


@pytest.mark.parametrize("T,n,din,dout,normalize", [(37, 32, 64, 64, False), (30, 10, 3, 50, True), (9, 50, 81, 256, True)])
def test_graphconv_fwd_bwd_and_bspmm(T, n, din, dout, normalize):
    rng = np.random.default_rng(T + din)
    adjs, (off, idx, val) = _batch(rng, T, n, normalize)
    x = rng.standard_normal((T, n, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal((1, dout)) * 0.3).astype(np.float32)
    g = rng.standard_normal((T, n, dout)).astype(np.float32)
    for nthreads in (1, 0):
        _close(ref_c.graphconv_fwd(off, idx, val, x, w, b, nthreads), K.graphconv_fwd(x, adjs, [w], [b]), 2e-6, "graphconv fwd")
        dx, dw, db = ref_c.graphconv_bwd(off, idx, val, x, w, g, nthreads)
        rdx, rdw, rdb = K.graphconv_bwd(x, adjs, [w], [b], g)
        _close(dx, rdx, 2e-6, "graphconv dX"); _close(dw, rdw[0], 3e-6, "graphconv dW"); _close(db, rdb[0], 3e-6, "graphconv dbias")
    rhs = rng.standard_normal((T, n, dout)).astype(np.float32)
    for adj_a in (False, True):
        ref = np.stack(K.bspmm([a[0] for a in adjs], list(rhs), adjoint_a=adj_a))
        _close(ref_c.bspmm(off, idx, val, rhs, n, n, adj_a), ref, 2e-6, "bspmm adjoint=%s" % adj_a)


def _act(v, act):
    return [v, 1 / (1 + np.exp(-v)), np.maximum(v, 0), np.tanh(v)][act]


def _dact(a, act):
    return [np.ones_like(a), a * (1 - a), (a > 0).astype(np.float64), 1 - a * a][act]


@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("m,din,dout", [(700, 256, 256), (333, 81, 256), (1000, 50, 50)])
def test_dense_fwd_bwd(m, din, dout, act):
    rng = np.random.default_rng(m + act)
    x = rng.standard_normal((m, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.2).astype(np.float32)
    g = rng.standard_normal((m, dout)).astype(np.float32)
    y64 = _act(x.astype(np.float64) @ w.astype(np.float64) + b, act)
    y = ref_c.dense_fwd(x, w, b, act)
    _close(y, y64, 2e-6, "dense fwd act=%d" % act)
    dx, dw, db = ref_c.dense_bwd(x, w, y, g, act)
    dpre = g.astype(np.float64) * _dact(y.astype(np.float64), act)        # the derivative is taken in the layer OUTPUT the C code was handed
    _close(dx, dpre @ w.astype(np.float64).T, 2e-6, "dense dX")
    _close(dw, x.astype(np.float64).T @ dpre, 2e-6, "dense dW")
    _close(db, dpre.sum(0), 2e-6, "dense dbias")
    dx0, _, _ = ref_c.dense_bwd(x, w, y, g, act, want_dx=False)
    assert dx0 is None


@pytest.mark.parametrize("T,n,d", [(40, 10, 256), (11, 50, 64)])
def test_gin_aggregate_with_adjoint_and_dot(T, n, d):
    rng = np.random.default_rng(T)
    adjs = K.synth_ring_graphs(rng, T, n) if n == 10 else K.synth_mol_graphs(rng, T, n, 3, normalize=True)
    off, idx, val = ref_c.flatten_coo([a[0] for a in adjs])
    x = rng.standard_normal((T, n, d)).astype(np.float32)
    g = rng.standard_normal((T, n, d)).astype(np.float32)
    eps = 0.37
    _close(ref_c.gin_aggregate(off, idx, val, x, eps), K.gin_fwd(x, adjs, [eps]), 2e-6, "GIN aggregate")
    dx, deps = K.gin_bwd(x, adjs, [eps], g)
    out, dot = ref_c.gin_aggregate(off, idx, val, g, eps, adjoint=True, dot_with=x)
    _close(out, dx, 2e-6, "GIN adjoint")
    assert abs(dot - deps[0]) <= 1e-9 * max(1.0, abs(deps[0])) + 1e-6 * np.sqrt(x.size) * 1e-3
