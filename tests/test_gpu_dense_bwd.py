"""The one-pass backward of the wide dense layers (kgcn_dense_bwd_f32, csrc/gemmb.hip) against the fp64 restatement of
kgcn/layers.py:248,260 (Keras Dense inside GraphDense) / :99-100, :112 (the X.W part of GraphConv) and their TF gradients:

    dpre = (grad [+ pooled_grad[row / n_nodes]]) (.) act'(act_out);   dx = dpre W^T;   dW = x^T dpre;   dbias = colsum dpre

through the raw C ABI (every activation x every gradient form, ragged last stage, padded leading dimensions, zero rows, row and
column magnitude spreads -- the kernel scales dpre per ROW and counter-scales x, see the file header) and through autograd
(ops.dense / ops.dense_gather take the one-pass route for wide layers that hand a gradient on); plus the old two-kernel route
on the same inputs as a second fp32 implementation."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import kgcn_oracle as K
from test_gpu_parity import close, dev, t32

pytestmark = pytest.mark.gpu
ACTS = {None: 0, "sigmoid": 1, "relu": 2, "tanh": 3}


def _act64(v, act):
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))
    if act == "relu":
        return np.maximum(v, 0)
    if act == "tanh":
        return np.tanh(v)
    return v


def _dact64(a, act):
    if act == "sigmoid":
        return a * (1 - a)
    if act == "relu":
        return (a > 0).astype(np.float64)
    if act == "tanh":
        return 1 - a * a
    return np.ones_like(a)


def _call(x, g, a, w, act, pooled=None, n_nodes=0, ld_pad=0, want_bias=True):
    """kgcn_dense_bwd_f32 on device copies of the operands (rows optionally padded by ld_pad floats) -> dx, dw, db (numpy)."""
    from kgcn_amd import _lib
    lib = _lib.lib
    m, din = x.shape
    dout = w.shape[1]

    def padded(arr):
        if arr is None:
            return None, 0
        buf = torch.full((arr.shape[0], arr.shape[1] + ld_pad), 7.5, device=dev(), dtype=torch.float32)
        buf[:, :arr.shape[1]] = t32(arr)
        return buf, arr.shape[1] + ld_pad

    tx, x_ld = padded(x)
    tg, ld = padded(g)
    ta, ld_a = padded(a)
    if tg is None:
        ld = ld_a
    assert ta is None or tg is None or ld == ld_a
    tw = t32(w)
    dx = torch.full((m, din + ld_pad), -3.25, device=dev(), dtype=torch.float32)
    dw = torch.empty((din, dout), device=dev(), dtype=torch.float32)
    db = torch.empty((dout,), device=dev(), dtype=torch.float32) if want_bias else None
    tb = int(lib.kgcn_dense_fwd_workspace_bytes(dout, din))
    tab = torch.empty((tb // 4,), device=dev(), dtype=torch.float32)
    wsb = int(lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout))
    ws = torch.empty((wsb // 4,), device=dev(), dtype=torch.float32)
    tp = None if pooled is None else t32(pooled)
    _lib.check(lib.kgcn_dense_bwd_f32(_lib.ptr(tg), _lib.ptr(tp), 0 if tp is None else tp.shape[1], n_nodes, _lib.ptr(ta),
                                      ACTS[act], ld, _lib.ptr(tx), x_ld, m, din, dout, _lib.ptr(tw), dout, _lib.ptr(dx),
                                      din + ld_pad, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(tab), tb, 0, _lib.ptr(ws), wsb,
                                      _lib.current_stream()), "kgcn_dense_bwd_f32")
    torch.cuda.synchronize()
    if ld_pad:
        assert bool((dx[:, din:] == -3.25).all()), "dx written beyond its columns"
    return dx[:, :din].cpu().numpy(), dw.cpu().numpy(), None if db is None else db.cpu().numpy()


def _reference(x, g, a, w, act, pooled=None, n_nodes=0):
    g64 = 0.0 if g is None else g.astype(np.float64)
    if pooled is not None:
        g64 = g64 + np.repeat(pooled.astype(np.float64), n_nodes, axis=0)
    dpre = g64 * (_dact64(a.astype(np.float64), act) if act else 1.0)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    return dpre, dpre @ w64.T, x64.T @ dpre, dpre.sum(0)


def _check(x, g, a, w, act, got, pooled=None, n_nodes=0, what="", tol_rows=2e-6, tol_w=3e-6):
    dpre, dx, dw, db = _reference(x, g, a, w, act, pooled, n_nodes)
    gdx, gdw, gdb = got
    # dX row r against the row's own scale sum_k |dpre[r, k]| |W[n, k]| (rows differ by many decades in the spread cases)
    scale = np.abs(dpre) @ np.abs(w.astype(np.float64)).T
    scale = np.maximum(scale.max(axis=1, keepdims=True), 1e-300)
    import conftest
    err = float((np.abs(gdx - dx) / scale).max())
    conftest.record_accuracy("%s dX / row scale" % what, err, tol_rows, 1.0)
    assert err <= tol_rows, (what, "dX", err)
    close(gdw, dw, atol=0, rel=tol_w, what="%s dW" % what)
    if gdb is not None:
        close(gdb, db, atol=0, rel=tol_w, what="%s dbias" % what)


def _layer(rng, m, din, dout, act):
    x = rng.standard_normal((m, din)).astype(np.float32)
    g = rng.standard_normal((m, dout)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    a = _act64(x.astype(np.float64) @ w.astype(np.float64) + b, act).astype(np.float32) if act else None
    return x, g, a, w


@pytest.mark.parametrize("act", [None, "sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("m,din,dout,ld_pad", [(16500, 256, 256, 0), (16384 + 31, 256, 256, 4), (16500, 192, 256, 0),
                                               (40000, 256, 256, 0), (16500, 132, 256, 0), (16411, 200, 256, 4)])
def test_one_pass_backward_against_fp64(act, m, din, dout, ld_pad):
    """every activation code; ragged last stage (m % 32 != 0), fewer stages than two per workgroup pair, padded leading
    dimensions (nothing outside the [m, din] block of dx may be touched), input widths below 256 (clamped column tiles; 132
    and 200 are what the aggregate-first route passes: F + 1 rounded up to a multiple of 4, partial last column tile)."""
    rng = np.random.default_rng(m + din + dout + len(str(act)))
    x, g, a, w = _layer(rng, m, din, dout, act)
    got = _call(x, g, a, w, act, ld_pad=ld_pad)
    _check(x, g, a, w, act, got, what="one-pass %s %d %d->%d" % (act, m, din, dout))


@pytest.mark.parametrize("act,with_rows", [("relu", True), ("relu", False), ("sigmoid", True), ("sigmoid", False),
                                           ("tanh", False)])
def test_one_pass_backward_with_the_read_out_gradient(act, with_rows):
    """the layer output was read out by GraphGather (example_model/model_gin.py:45-60): the gradient of node row r is
    grad[r] + d pooled[r / N] -- or the broadcast alone when the output was not passed on (grad = NULL)."""
    rng = np.random.default_rng(77 + with_rows)
    N, T, din, dout = 10, 1700, 256, 256
    m = N * T
    x, g, a, w = _layer(rng, m, din, dout, act)
    pooled = rng.standard_normal((T, dout)).astype(np.float32)
    got = _call(x, g if with_rows else None, a, w, act, pooled=pooled, n_nodes=N)
    _check(x, g if with_rows else None, a, w, act, got, pooled, N, what="one-pass %s pooled%s" % (act, "+rows" if with_rows else ""))


@pytest.mark.parametrize("case", ["zero_rows", "row_exponents", "column_exponents", "heavy_tail", "denormals"])
def test_one_pass_backward_scaling_edges(case):
    """The kernel's arithmetic is f16 x 2 with a ROW scale on d pre-activation and a counter-scaled, per-column online scale
    on x (gemmb.hip header).  zero_rows: the padding rows of a ragged-compact batch (all-zero gradient rows, arbitrary x) must
    neither contribute nor set a column scale; row_exponents: gradient rows spread over 18 decades, x rows over 12;
    column_exponents: x columns spread over 12 decades (the online column scale must follow: accumulator rescales mid-launch, the
    large rows come LAST); heavy_tail: Cauchy-distributed gradients; denormals: fp32-denormal gradient rows."""
    rng = np.random.default_rng(sum(map(ord, case)))
    m, din, dout = 20000, 256, 256
    x, g, a, w = _layer(rng, m, din, dout, "relu")
    tol_rows, tol_w = 2e-6, 3e-6
    if case == "zero_rows":
        g[5000:] = 0                                         # three quarters of the rows carry no gradient
        x[5000:] *= 1e6                                      # ... and would dominate every column scale of x if they counted
    elif case == "row_exponents":
        g *= (10.0 ** rng.uniform(-9, 9, size=(m, 1))).astype(np.float32)
        x *= (10.0 ** rng.uniform(-6, 6, size=(m, 1))).astype(np.float32)
    elif case == "column_exponents":
        x *= (10.0 ** rng.uniform(-6, 6, size=(1, din))).astype(np.float32)
        order = np.argsort(np.abs(g).max(axis=1))            # rows with the largest gradients at the END of the batch
        x, g, a = x[order], g[order], a[order]
        x[-64:] *= 4096.0
    elif case == "heavy_tail":
        g = rng.standard_cauchy((m, dout)).astype(np.float32)
    elif case == "denormals":
        g[::2] = (g[::2] * np.float32(2.0 ** -140)).astype(np.float32)
    got = _call(x, g, a, w, "relu")
    dpre, dx, dw, db = _reference(x, g, a, w, "relu")
    gdx, gdw, gdb = got
    assert np.isfinite(gdx).all() and np.isfinite(gdw).all()
    import conftest
    scale = np.maximum((np.abs(dpre) @ np.abs(w.astype(np.float64)).T).max(axis=1, keepdims=True), 1e-300)
    # (a result that is itself an fp32 denormal is exact only to the denormal spacing 2^-149: eight of them are granted)
    err = float((np.maximum(np.abs(gdx - dx) - 8 * 2.0 ** -149, 0.0) / scale).max())
    conftest.record_accuracy("%s dX / row scale" % case, err, tol_rows, 1.0)
    assert err <= tol_rows, (case, "dX", err)
    # dW[i, j] against ITS natural scale sum_r |x[r, i]| |dpre[r, j]| (columns of x differ by 12 decades in one case)
    wscale = np.maximum(np.abs(x.astype(np.float64)).T @ np.abs(dpre), 1e-300)
    errw = float((np.abs(gdw - dw) / wscale).max())
    np32 = (x.T @ (dpre.astype(np.float32))).astype(np.float64)          # numpy float32 on the same operands
    errn = float((np.abs(np32 - dw) / wscale).max())
    conftest.record_accuracy("%s dW / sum|x||dpre| (numpy float32: %.2e)" % (case, errn), errw, max(4 * errn, 1e-6), 1.0)
    assert errw <= max(4 * errn, 1e-6), (case, "dW", errw, errn)
    close(gdb, db, atol=0, rel=tol_w, what="%s dbias" % case)


def test_one_pass_backward_propagates_non_finite_values_and_nothing_else():
    rng = np.random.default_rng(5)
    m, din, dout = 16500, 256, 256
    x, g, a, w = _layer(rng, m, din, dout, "tanh")
    g[100, 7] = np.inf
    g[100, 8:40] *= 50.0                                      # finite neighbours of the inf, large enough to overflow f16 at a careless row scale
    x[9000, 33] = np.nan
    gdx, gdw, gdb = _call(x, g, a, w, "tanh")
    assert not np.isfinite(gdx[100]).any(), "row 100 of dx depends on the inf in grad[100]"
    ok_rows = np.ones(m, bool); ok_rows[100] = False
    assert np.isfinite(gdx[ok_rows]).all()
    assert not np.isfinite(gdw[:, 7]).any() and not np.isfinite(gdw[33, :]).any()
    mask = np.ones((din, dout), bool); mask[:, 7] = False; mask[33, :] = False
    assert np.isfinite(gdw[mask]).all()
    assert not np.isfinite(gdb[7]) and np.isfinite(np.delete(gdb, 7)).all()


def test_one_pass_backward_argument_checks():
    from kgcn_amd import _lib
    lib = _lib.lib
    assert lib.kgcn_dense_bwd_supported(16384, 256, 256) == 1 and lib.kgcn_dense_bwd_supported(16383, 256, 256) == 0
    assert lib.kgcn_dense_bwd_supported(20000, 128, 256) == 0 and lib.kgcn_dense_bwd_supported(20000, 256, 252) == 0
    z = ctypes.c_void_p(0)
    p = ctypes.c_void_p(4096)
    # unsupported shape, NULL operands, pooled gradient without an activation, dx aliasing an input: status + message, no launch
    assert lib.kgcn_dense_bwd_f32(p, z, 0, 0, p, 2, 256, p, 256, 100, 256, 256, p, 256, p, 256, p, p, p, 1 << 30, 1, p, 1 << 30, z) != 0
    assert b"one-pass" in lib.kgcn_last_error()
    assert lib.kgcn_dense_bwd_f32(z, z, 0, 0, p, 2, 256, p, 256, 20000, 256, 256, p, 256, p, 256, p, p, p, 1 << 30, 1, p, 1 << 30, z) != 0
    assert b"NULL" in lib.kgcn_last_error()
    assert lib.kgcn_dense_bwd_f32(p, p, 256, 10, z, 0, 256, p, 256, 20000, 256, 256, p, 256, p, 256, p, p, p, 1 << 30, 1, p, 1 << 30, z) != 0
    assert b"pooled" in lib.kgcn_last_error()
    q = ctypes.c_void_p(8192)
    assert lib.kgcn_dense_bwd_f32(p, z, 0, 0, p, 2, 256, q, 256, 20000, 256, 256, p, 256, q, 256, p, p, p, 1 << 30, 1, p, 1 << 30, z) != 0
    assert b"alias" in lib.kgcn_last_error()
    r = ctypes.c_void_p(12288)
    assert lib.kgcn_dense_bwd_f32(p, z, 0, 0, p, 2, 256, q, 256, 20000, 256, 256, p, 256, r, 256, p, p, p, 1 << 30, 1, p, 16, z) != 0
    assert b"workspace" in lib.kgcn_last_error()


@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
def test_autograd_takes_the_one_pass_route_and_agrees_with_the_two_kernel_route(act):
    """ops.dense on a wide layer that hands a gradient on: the backward is ONE C-ABI call (kgcn_dense_bwd_f32); the two-kernel
    route (kgcn_dense_dx_dact_f32 + kgcn_dense_wgrad_f32, ops.dense_bwd_fusion = False) on the same inputs is the second fp32
    implementation -- both against fp64."""
    from kgcn_amd import ops
    rng = np.random.default_rng(31)
    m, din, dout = 18000, 256, 256
    x, g, a, w = _layer(rng, m, din, dout, act)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    res = {}
    for fused in (True, False):
        ops.dense_bwd_fusion = fused
        try:
            tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
            y = ops.dense(tx, tw, tb, activation=act)
            y.backward(t32(g))
            res[fused] = (tx.grad.cpu().numpy(), tw.grad.cpu().numpy(), tb.grad.cpu().numpy(), y.detach().cpu().numpy())
        finally:
            ops.dense_bwd_fusion = True
    a_dev = res[True][3] if act else None                      # the saved activation the backward really saw
    _check(x, g, a_dev, w, act, res[True][:3], what="autograd one-pass %s" % act)
    _check(x, g, a_dev, w, act, res[False][:3], what="autograd two-kernel %s" % act)
    for i, name in enumerate(("dX", "dW", "dbias")):
        close(res[True][i], res[False][i], atol=0, rel=4e-6, what="one-pass vs two-kernel %s (%s)" % (name, act))


def test_dense_gather_backward_takes_the_one_pass_route():
    """ops.dense_gather (GraphDense + GraphGather, model_gin.py:45-60): y handed on AND read out -> grad + pooled in one pass."""
    from kgcn_amd import ops
    rng = np.random.default_rng(41)
    T, N, din, dout = 1800, 10, 256, 256
    x = rng.standard_normal((T, N, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    gy = rng.standard_normal((T, N, dout)).astype(np.float32)
    gp = rng.standard_normal((T, dout)).astype(np.float32)
    res = {}
    for fused in (True, False):
        ops.dense_bwd_fusion = fused
        try:
            tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
            y, pooled = ops.dense_gather(tx, tw, tb, activation="relu")
            ((y * t32(gy)).sum() + (pooled * t32(gp)).sum()).backward()
            res[fused] = (tx.grad.reshape(T * N, din).cpu().numpy(), tw.grad.cpu().numpy(), tb.grad.cpu().numpy(),
                          y.detach().reshape(T * N, dout).cpu().numpy())
        finally:
            ops.dense_bwd_fusion = True
    a_dev = res[True][3]
    _check(x.reshape(T * N, din), gy.reshape(T * N, dout), a_dev, w, "relu", res[True][:3], gp, N, what="dense_gather one-pass")
    for i, name in enumerate(("dX", "dW", "dbias")):
        close(res[True][i], res[False][i], atol=0, rel=4e-6, what="dense_gather one-pass vs two-kernel %s" % name)
