"""Closable parity holes of round 2:

(a) the exact 3-way bf16 split under stress in EVERY kernel that uses it, not only the fused 32x64 layer: the wide-layer
    GEMMs gemm3 forward / dX / weight gradient (256 -> 256, 81 -> 256, with and without the W fragment table: the table is
    only used from 1,024 rows), gemmn (256 -> 50), wgradn (256 x 50), dense_dx_dact (sigmoid / relu / tanh riding in the dX
    GEMM, also with a layer input wider than 256: several column blocks, ONE of them writes d pre-activation).
    Inputs: per-row exponents 1e-18 .. 1e18, all-ones significands, fp32 denormals, +-inf / NaN.  References: the fp64
    oracle (numpy) and, as a second independent fp32 implementation, rocBLAS / hipBLASLt through torch.matmul.
(b) BASELINE configs 4 and 5 at their FULL benchmark sizes (204,800 / 200,000 node rows): forward, dX and the weight /
    bias gradients -- sums over 2e5 rows through up to 256 row-range partials -- against the C restatement
    (oracle/kgcn_ref.c, fp32 products, fp64 row sums), in the padded AND the ragged-compact layout.
"""
import numpy as np
import pytest
import torch

from oracle import kgcn_oracle as K
from test_gpu_parity import _row_close, close, dev, t32

pytestmark = pytest.mark.gpu

# (16,500 / 16,450 rows: the f16 two-piece kernels of gemmh.hip take the wide layers from 16,384 rows -- 256 tiles of 64 -- on;
#  below that the bf16 x 3 kernels of gemm3.hip run, with the weight table from 1,024 rows)
SHAPES = [(4100, 256, 256), (300, 256, 256), (4101, 81, 256), (4100, 256, 50), (1500, 300, 256), (4100, 512, 96),
          (16500, 256, 256), (16450, 84, 256), (16390, 128, 192)]
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


def _inputs(case, M, din, dout, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, din)).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    if case == "exponents":
        x *= (10.0 ** rng.uniform(-18, 18, size=(M, 1))).astype(np.float32)
        g *= (10.0 ** rng.uniform(-9, 9, size=(M, 1))).astype(np.float32)
        b[:] = 0                                   # a bias of 0.1 would swamp rows of scale 1e-18: the row test needs a scale
    elif case == "full_significands":
        ones = lambda a: (a.view(np.uint32) | np.uint32(0x007fffff)).view(np.float32)
        x, g, w = ones(x), ones(g), ones(w)
    elif case == "denormals":
        tiny = np.float32(2.0 ** -140)
        x[::2] = (x[::2] * tiny).astype(np.float32)
        assert (np.abs(x[::2]) < 2.0 ** -126).all() and (x[::2] != 0).any()
        b[:] = 0
    return x, g, w, b


@pytest.mark.parametrize("case", ["normal", "exponents", "full_significands", "denormals"])
@pytest.mark.parametrize("M,din,dout", SHAPES)
def test_bf16_split_dense_kernels_edge_values(case, M, din, dout):
    from kgcn_amd import ops
    x, g, w, b = _inputs(case, M, din, dout, seed=M + din + dout)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    y = ops.dense(tx, tw, tb)
    y.backward(t32(g))
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
    ref_y, ref_dx, ref_dw, ref_db = x64 @ w64 + b, g64 @ w64.T, x64.T @ g64, g64.sum(0)
    assert np.isfinite(ref_y).all() and np.isfinite(ref_dw).all()
    _row_close(y, ref_y, 2e-6, "%s dense fwd %d->%d" % (case, din, dout))
    _row_close(tx.grad, ref_dx, 2e-6, "%s dense dX" % case)
    if case == "exponents":            # dW rows mix row scales of 36 decades: meaningful against the row's own maximum only
        _row_close(tw.grad, ref_dw, 1e-5, "%s dense dW" % case)
    else:
        close(tw.grad, ref_dw, atol=0, rel=3e-6, what="%s dense dW" % case)
        close(tb.grad, ref_db, atol=0, rel=3e-6, what="%s dense dbias" % case)
    # second fp32 implementation (rocBLAS / hipBLASLt through torch): same inputs, agreement to a few fp32 roundings
    if case != "denormals":            # the BLAS kernels flush fp32 denormals; ours keep them (checked against fp64 above)
        ty = torch.matmul(t32(x), t32(w)) + t32(b)
        _row_close(y, ty.cpu().numpy(), 4e-6, "%s dense fwd vs torch.matmul" % case)
        _row_close(tx.grad, torch.matmul(t32(g), t32(w).t()).cpu().numpy(), 4e-6, "%s dense dX vs torch.matmul" % case)


@pytest.mark.parametrize("act", ["sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("M,din,dout", [(4100, 256, 256), (4101, 300, 256), (333, 256, 256), (4100, 256, 50), (2050, 81, 256),
                                        (16500, 256, 256), (16401, 256, 192)])
def test_dense_dx_dact_all_routes(act, M, din, dout):
    """act(x W + b) with the activation in the GEMM epilogue and its derivative riding in the dX GEMM (kgcn_dense_dx_dact_f32);
    din = 300 > 256: two column blocks stage the same gradient rows, only blockIdx.y == 0 stores d pre-activation
    (gemm3.hip) -- dW / dbias below are computed FROM that stored tensor."""
    from kgcn_amd import ops
    rng = np.random.default_rng(M + din)
    x = rng.standard_normal((M, din)).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    y = ops.dense(tx, tw, tb, activation=act)
    y.backward(t32(g))
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    a = _act64(x64 @ w64 + b, act)
    if act == "relu":                   # the mask comes from the GPU activations (a pre-activation of 1e-9 may flip its sign)
        dpre = g.astype(np.float64) * (y.detach().cpu().numpy() > 0)
    else:
        dpre = g.astype(np.float64) * _dact64(a, act)
    close(y, a, atol=1e-6, rel=1e-6, what="act(dense) fwd")
    close(tx.grad, dpre @ w64.T, atol=0, rel=3e-6, what="dX with d activation")
    close(tw.grad, x64.T @ dpre, atol=0, rel=3e-6, what="dW from the stored d pre-activation")
    close(tb.grad, dpre.sum(0), atol=0, rel=3e-6, what="dbias from the stored d pre-activation")
    # the stored d pre-activation against the stand-alone activation backward kernel
    ref_dpre = ops.activation_backward(y.detach(), t32(g), ACTS[act])
    close(ref_dpre, dpre, atol=0, rel=2e-6, what="kgcn_act_bwd_f32")


@pytest.mark.parametrize("act", ["sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("M,din,dout", [(4100, 256, 256), (4101, 84, 256), (333, 100, 300), (2050, 300, 256), (16500, 256, 256)])
def test_dense_wgrad_with_fused_activation_derivative(act, M, din, dout):
    """act(x W + b) whose INPUT needs no gradient (the first layer of a model): d pre-activation is formed inside the wide
    weight-gradient GEMM's staging (kgcn_dense_wgrad_dact_f32) -- dW / dbias vs fp64 and vs the unfused route (activation backward
    pass + plain weight gradient)."""
    from kgcn_amd import ops
    rng = np.random.default_rng(M + dout)
    x = rng.standard_normal((M, din)).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    res = {}
    for fused in (True, False):
        ops.wgrad_dact_fusion = fused
        try:
            tw, tb = t32(w).requires_grad_(True), t32(b).requires_grad_(True)
            y = ops.dense(t32(x), tw, tb, activation=act)
            y.backward(t32(g))
            res[fused] = (y.detach().cpu().numpy(), tw.grad.cpu().numpy(), tb.grad.cpu().numpy())
        finally:
            ops.wgrad_dact_fusion = True
    yo = res[True][0]
    a = _act64(x.astype(np.float64) @ w.astype(np.float64) + b, act)
    dpre = g.astype(np.float64) * ((yo > 0) if act == "relu" else _dact64(a, act))
    close(res[True][1], x.astype(np.float64).T @ dpre, atol=0, rel=3e-6, what="dW, d activation inside the weight-gradient GEMM")
    close(res[True][2], dpre.sum(0), atol=0, rel=3e-6, what="dbias, d activation inside the weight-gradient GEMM")
    close(res[True][1], res[False][1], atol=0, rel=3e-6, what="fused vs unfused dW")
    close(res[True][2], res[False][2], atol=0, rel=3e-6, what="fused vs unfused dbias")


@pytest.mark.parametrize("M,din,dout", [(4100, 256, 256), (300, 256, 256), (4100, 81, 256), (4100, 256, 50), (16500, 256, 256),
                                        (16450, 84, 256)])
def test_bf16_split_dense_kernels_non_finite(M, din, dout):
    """+-inf splits into (inf, NaN, NaN), NaN into three NaNs: every output element that depends on a non-finite input is
    non-finite, every other element is untouched (include/kgcn_hip.h)."""
    from kgcn_amd import ops
    rng = np.random.default_rng(7)
    x = rng.standard_normal((M, din)).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    bad = [(5, 3, np.inf), (77, din - 1, -np.inf), (M - 1, 0, np.nan)]
    xb = x.copy()
    for r, k, v in bad:
        xb[r, k] = v
    tx, tw = t32(xb).requires_grad_(True), t32(w).requires_grad_(True)
    y = ops.dense(tx, tw, None)
    y.backward(t32(g))
    yo = y.detach().cpu().numpy()
    rows = [r for r, _, _ in bad]
    assert not np.isfinite(yo[rows]).any()
    clean = np.ones(M, bool); clean[rows] = False
    np.testing.assert_allclose(yo[clean], (x.astype(np.float64) @ w)[clean], rtol=0, atol=2e-5)
    np.testing.assert_allclose(tx.grad.cpu().numpy(), g.astype(np.float64) @ w.T.astype(np.float64), rtol=0, atol=2e-5)
    dw = tw.grad.cpu().numpy()
    ks = sorted({k for _, k, _ in bad})
    assert not np.isfinite(dw[ks]).any() and np.isfinite(np.delete(dw, ks, axis=0)).all()


# ---------------------------------------------------------------------------------------------------------------------
# (b) BASELINE configs 4 / 5 at full benchmark size vs the C restatement
# ---------------------------------------------------------------------------------------------------------------------
def _cfg4_batch(B=4096, N=50, F=81):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sizes, g, r, c, rng = bench.gen_tox21_like(B, N, seed=4)
    val = bench.kipf_values(g, r, c, B, N)
    valid = np.arange(N)[None, :] < sizes[:, None]
    x = rng.standard_normal((B, N, F)).astype(np.float32) * valid[:, :, None]
    off = np.zeros(B + 1, np.int64)
    np.cumsum(np.bincount(g, minlength=B), out=off[1:])
    idx = np.ascontiguousarray(np.stack([r, c], 1).astype(np.int32))
    return sizes, g, r, c, val, x, off, idx, rng


def test_cfg4_full_size_layers_vs_c_oracle():
    """4,096 Tox21-shaped molecules x 50 padded nodes = 204,800 rows: GraphConv 81 -> 256 (sigmoid), GraphDense 256 -> 256
    (sigmoid), GraphConv 256 -> 50, GraphDense 50 -> 50 (sigmoid), each forward + backward on the padded layout AND on the
    ragged-compact layout, against oracle/kgcn_ref.c."""
    from kgcn_amd import BatchedAdjacency, BatchedCSR, layers, ragged
    from oracle import ref_c
    B, N, F = 4096, 50, 81
    sizes, g, r, c, val, x, off, idx, rng = _cfg4_batch(B, N, F)
    csr = BatchedCSR.from_arrays(g, r, c, val, B, N, N, device=dev())
    adj = BatchedAdjacency([csr])
    rb = ragged.compact(t32(x), adj, sizes)
    valid = (np.arange(N)[None, :] < sizes[:, None])
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))

    def run(layer, inp, upstream, ragged_mode, **kw):
        """-> (out, d input, parameter grads), everything brought back to the padded layout."""
        if ragged_mode:
            ti = rb.compact_rows(t32(inp)).requires_grad_(True)
            out = layer(ti, adj=rb) if kw.get("conv") else layer(ti)
            out.backward(rb.compact_rows(t32(upstream)))
            return rb.expand(out.detach(), fill="zero").cpu().numpy(), rb.expand(ti.grad, fill="zero").cpu().numpy()
        ti = t32(inp).requires_grad_(True)
        out = layer(ti, adj=adj) if kw.get("conv") else layer(ti)
        out.backward(t32(upstream))
        return out.detach().cpu().numpy(), ti.grad.cpu().numpy()

    # ---- GraphConv din -> dout with sigmoid ----
    for din, dout, act in ((F, 256, "sigmoid"), (256, 50, None)):
        inp = x if din == F else (rng.standard_normal((B, N, din)).astype(np.float32) * valid[:, :, None])
        up = rng.standard_normal((B, N, dout)).astype(np.float32) * valid[:, :, None]
        for mode in (False, True):
            torch.manual_seed(1)
            layer = layers.GraphConv(dout, 1, activation=act)
            layer.build((B, N, din), dev())
            with torch.no_grad():
                layer.bias[0].copy_(t32(rng.standard_normal((1, dout)) * 0.1))
            wn, bn = layer.w[0].detach().cpu().numpy(), layer.bias[0].detach().cpu().numpy()
            out, dinp = run(layer, inp, up, mode, conv=True)
            pre = ref_c.graphconv_fwd(off, idx, val, inp, wn, bn)
            ref_out = sig(pre.astype(np.float64)) if act else pre
            if mode:                                   # ragged: padded rows are not stored (they are act(0) constants)
                ref_cmp, out_cmp = ref_out * valid[:, :, None], out * valid[:, :, None]
            else:
                ref_cmp, out_cmp = ref_out, out
            close(out_cmp, ref_cmp, atol=1e-6, rel=2e-6, what="cfg4 full GraphConv %d->%d fwd (ragged=%s)" % (din, dout, mode))
            gpre = (up * (ref_out * (1 - ref_out) if act else 1.0)).astype(np.float32)
            rdx, rdw, rdb = ref_c.graphconv_bwd(off, idx, val, inp, wn, gpre)
            close(dinp, rdx, atol=0, rel=5e-6, what="cfg4 full GraphConv %d->%d dX (ragged=%s)" % (din, dout, mode))
            close(layer.w[0].grad, rdw, atol=0, rel=2e-5, what="cfg4 full GraphConv %d->%d dW (ragged=%s)" % (din, dout, mode))
            close(layer.bias[0].grad, rdb, atol=0, rel=2e-5, what="cfg4 full GraphConv %d->%d dbias (ragged=%s)" % (din, dout, mode))
    # ---- GraphDense din -> dout with sigmoid (all 204,800 rows; ragged: the valid rows + the padding representative) ----
    for din, dout in ((256, 256), (50, 50)):
        inp = rng.standard_normal((B, N, din)).astype(np.float32)
        up = rng.standard_normal((B, N, dout)).astype(np.float32)
        torch.manual_seed(2)
        layer = layers.GraphDense(dout, activation="sigmoid")
        layer.build((B, N, din), dev())
        with torch.no_grad():
            layer.bias.copy_(t32(rng.standard_normal(dout) * 0.1))
        kn, bn = layer.kernel.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
        out, dinp = run(layer, inp, up, False)
        x2, u2 = inp.reshape(B * N, din), up.reshape(B * N, dout)
        ry = ref_c.dense_fwd(x2, kn, bn, act=1)
        close(out.reshape(B * N, dout), ry, atol=1e-6, rel=2e-6, what="cfg4 full GraphDense %d->%d fwd" % (din, dout))
        rdx, rdw, rdb = ref_c.dense_bwd(x2, kn, ry, u2, act=1)
        close(dinp.reshape(B * N, din), rdx, atol=0, rel=5e-6, what="cfg4 full GraphDense dX")
        close(layer.kernel.grad, rdw, atol=0, rel=5e-6, what="cfg4 full GraphDense %d->%d dK over 204,800 rows" % (din, dout))
        close(layer.bias.grad, rdb, atol=0, rel=5e-6, what="cfg4 full GraphDense %d->%d dbias over 204,800 rows" % (din, dout))


def test_cfg5_full_size_layers_vs_c_oracle():
    """20,000 ring graphs x 10 nodes x 256 features = 200,000 rows: GINAggregate (epsilon = 0.25) forward, d x, d epsilon and
    GraphDense 256 -> 256 (relu) forward, dX, dK, dbias against oracle/kgcn_ref.c."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from kgcn_amd import BatchedAdjacency, BatchedCSR, layers
    from oracle import ref_c
    B, N, D = 20000, 10, 256
    g, r, c, _, rng = bench.gen_ring_graphs(B, N, seed=5)
    val = np.ones(g.shape[0], np.float32)
    off = np.zeros(B + 1, np.int64)
    np.cumsum(np.bincount(g, minlength=B), out=off[1:])
    idx = np.ascontiguousarray(np.stack([r, c], 1).astype(np.int32))
    adj = BatchedAdjacency([BatchedCSR.from_arrays(g, r, c, val, B, N, N, device=dev())])
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    up = rng.standard_normal((B, N, D)).astype(np.float32)
    gin = layers.GINAggregate(1)
    tx = t32(x).requires_grad_(True)
    gin(tx, adj=adj)
    with torch.no_grad():
        gin.epsilon[0].fill_(0.25)
    out = gin(tx, adj=adj)
    out.backward(t32(up))
    ro = ref_c.gin_aggregate(off, idx, val, x, 0.25)
    close(out, ro, atol=0, rel=2e-6, what="cfg5 full GINAggregate fwd")
    rdx, deps = ref_c.gin_aggregate(off, idx, val, up, 0.25, adjoint=True, dot_with=x)
    close(tx.grad, rdx, atol=0, rel=2e-6, what="cfg5 full GINAggregate dX")
    scale = float(np.abs(up.astype(np.float64) * x).sum())
    assert abs(float(gin.epsilon[0].grad) - deps) <= 2e-6 * scale, (float(gin.epsilon[0].grad), deps, scale)
    layer = layers.GraphDense(D, activation="relu")
    layer.build((B, N, D), dev())
    with torch.no_grad():
        layer.bias.copy_(t32(rng.standard_normal(D) * 0.1))
    kn, bn = layer.kernel.detach().cpu().numpy(), layer.bias.detach().cpu().numpy()
    ti = t32(x).requires_grad_(True)
    y = layer(ti)
    y.backward(t32(up))
    x2, u2 = x.reshape(B * N, D), up.reshape(B * N, D)
    yo = y.detach().cpu().numpy().reshape(B * N, D)
    ry = ref_c.dense_fwd(x2, kn, bn, act=2)
    close(yo, ry, atol=1e-6, rel=2e-6, what="cfg5 full GraphDense fwd")
    # relu mask from the GPU activations (a pre-activation of 1e-8 may flip its sign between two fp32 summation orders)
    rdx, rdw, rdb = ref_c.dense_bwd(x2, kn, yo, u2, act=2)
    close(ti.grad.reshape(B * N, D), rdx, atol=0, rel=5e-6, what="cfg5 full GraphDense dX")
    close(layer.kernel.grad, rdw, atol=0, rel=5e-6, what="cfg5 full GraphDense dK over 200,000 rows")
    close(layer.bias.grad, rdb, atol=0, rel=5e-6, what="cfg5 full GraphDense dbias over 200,000 rows")


@pytest.mark.parametrize("tee", [True, False])
@pytest.mark.parametrize("act,T,N,din,dout", [("relu", 500, 10, 256, 256), ("sigmoid", 130, 32, 300, 256), ("relu", 60, 10, 50, 50),
                                               (None, 200, 7, 256, 256), ("tanh", 512, 4, 256, 512),
                                               # whole rounds of 64-row tiles + a second, narrow-block launch over rows 32,768..:
                                               # the graph of a row there needs the absolute row index (G3Dact.row0)
                                               ("relu", 3616, 10, 256, 256), ("sigmoid", 5200, 7, 256, 256)])
def test_dense_gather_gradient_joins_inside_the_dx_gemm(act, T, N, din, dout, tee):
    """ops.dense_gather = GraphDense (+ activation) followed by GraphGather, the layer output optionally handed on as well
    (model_gin.py:45-60).  Backward of the wide activated cases goes through kgcn_dense_dx_dact_gather_f32 (d pooled's broadcast
    formed inside the dX GEMM's staging; `tee`: added to the passed-on gradient there), the others through the fallback; all
    against an fp64 evaluation: outputs, d inputs, dW, dbias."""
    from kgcn_amd import ops
    rng = np.random.default_rng(T + N + dout)
    x = rng.standard_normal((T, N, din)).astype(np.float32)
    w = (rng.standard_normal((din, dout)) / np.sqrt(din)).astype(np.float32)
    b = rng.standard_normal(dout).astype(np.float32)
    gp = rng.standard_normal((T, dout)).astype(np.float32)
    gy = rng.standard_normal((T, N, dout)).astype(np.float32)
    tx, tw, tb = (t32(a).requires_grad_(True) for a in (x, w, b))
    y, pooled = ops.dense_gather(tx, tw, tb, activation=act)
    loss = (pooled * t32(gp)).sum() + ((y * t32(gy)).sum() if tee else 0.0)
    loss.backward()
    pre = x.astype(np.float64).reshape(T * N, din) @ w.astype(np.float64) + b
    f = {"relu": lambda v: np.maximum(v, 0), "sigmoid": lambda v: 1 / (1 + np.exp(-v)), "tanh": np.tanh, None: lambda v: v}[act]
    df = {"relu": lambda a: (a > 0).astype(np.float64), "sigmoid": lambda a: a * (1 - a), "tanh": lambda a: 1 - a * a,
          None: lambda a: np.ones_like(a)}[act]
    yr = f(pre)
    g = np.repeat(gp.astype(np.float64), N, axis=0) + (gy.reshape(T * N, dout) if tee else 0.0)
    # relu: the mask of the GPU's own activations (among millions of pre-activations one within an fp32 rounding of zero
    # changes its sign between two summation orders)
    dpre = g * (df(yr) if act != "relu" else (y.detach().cpu().numpy().reshape(T * N, dout) > 0))
    scale = float(np.abs(yr).max())
    close(y, yr.reshape(T, N, dout), atol=2e-6 * max(1.0, scale), rel=2e-6, what="dense_gather y")
    close(pooled, yr.reshape(T, N, dout).sum(1), atol=2e-5 * max(1.0, scale), rel=2e-6, what="dense_gather pooled")
    close(tx.grad, (dpre @ w.astype(np.float64).T).reshape(T, N, din), atol=1e-5, rel=2e-5, what="dense_gather d inputs")
    close(tw.grad, x.astype(np.float64).reshape(T * N, din).T @ dpre, atol=1e-4, rel=2e-5, what="dense_gather dW")
    close(tb.grad, dpre.sum(0), atol=1e-4, rel=2e-5, what="dense_gather dbias")


@pytest.mark.parametrize("act,T,N,d", [("relu", 500, 10, 256), ("sigmoid", 130, 32, 256), ("relu", 60, 10, 52)])
def test_read_outs_joined_in_one_buffer_equal_concatenation(act, T, N, d):
    """model_gin.py:61 concatenates the read-outs of its blocks.  ops.dense_gather(join=buf, join_col=..) writes each read-out into its
    column block of one buffer and ops.join_columns hands the buffer on: the same values as torch.cat, and the gradient of every
    block -- a strided column block of d buffer, read in place by kgcn_dense_dx_dact_gather_f32 -- the same bits as through the
    concatenation (same kernels, same operands; only the pooled gradient's row stride differs)."""
    from kgcn_amd import ops
    rng = np.random.default_rng(T + d)
    xs = [rng.standard_normal((T, N, d)).astype(np.float32) for _ in range(2)]
    ws = [(rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32) for _ in range(2)]
    bs = [rng.standard_normal(d).astype(np.float32) for _ in range(2)]
    head = t32(rng.standard_normal((2 * d, 3)))
    res = []
    for joined in (False, True):
        tx = [t32(a).requires_grad_(True) for a in xs]
        tw = [t32(a).requires_grad_(True) for a in ws]
        tb = [t32(a).requires_grad_(True) for a in bs]
        buf = torch.empty((T, 2 * d), device=dev()) if joined else None
        parts, ys = [], []
        for i in range(2):
            y, p = ops.dense_gather(tx[i], tw[i], tb[i], activation=act, join=buf, join_col=i * d)
            parts.append(p); ys.append(y)
        cat = ops.join_columns(buf, parts) if joined else torch.cat(parts, dim=1)
        loss = (torch.tanh(cat @ head)).sum() + (ys[0] * 0.5).sum()          # block 0 is handed on as well
        loss.backward()
        res.append([cat.detach().clone()] + [t.grad.clone() for t in tx + tw + tb])
    for a, b in zip(*res):
        assert torch.equal(a, b), float((a - b).abs().max())
    with pytest.raises(Exception, match="join"):
        ops.join_columns(torch.empty((T, 2 * d), device=dev()), [res[0][0][:, :d], res[0][0][:, d:]])


@pytest.mark.parametrize("act,T,N,d,dout", [("relu", 2000, 10, 256, 256), ("sigmoid", 1700, 10, 256, 256), ("relu", 640, 32, 160, 256),
                                             ("relu", 300, 10, 256, 256)])
def test_gin_dense_d_epsilon_inside_the_dx_gemm(act, T, N, d, dout):
    """ops.gin_dense = GINAggregate + activated GraphDense (model_gin.py:45-50).  With inputs that need no gradient d epsilon =
    <d out, x> (kgcn/layers.py:469) is accumulated by the dX GEMM (kgcn_dense_dx_dact_dot_f32: the product is never stored); the last
    case is below the fused form's size and takes the two ops.  Checked against the two separate ops (y bit-equal, dW / dbias to
    fp32 rounding) and against fp64 for y and d epsilon."""
    from kgcn_amd import ops
    from kgcn_amd.batched_csr import BatchedAdjacency
    from oracle import kgcn_oracle as K
    rng = np.random.default_rng(T + d)
    adjs = K.synth_mol_graphs(rng, T, N, 1)
    a = BatchedAdjacency.from_adjs(adjs, device=dev())
    x = rng.standard_normal((T, N, d)).astype(np.float32)
    w = (rng.standard_normal((d, dout)) / np.sqrt(d)).astype(np.float32)
    b = (rng.standard_normal(dout) * 0.1).astype(np.float32)
    gy = rng.standard_normal((T, N, dout)).astype(np.float32)
    res = []
    for fused in (True, False):
        tx = t32(x).requires_grad_(not fused)                      # an input that needs a gradient takes the two separate ops
        eps = t32(np.array([0.3])).requires_grad_(True)
        tw, tb = t32(w).requires_grad_(True), t32(b).requires_grad_(True)
        y = ops.gin_dense(tx, eps, a, tw, tb, activation=act)
        assert (y.grad_fn.name().startswith("_GinDense")) == (fused and T * N >= 16384)
        (y * t32(gy)).sum().backward()
        res.append((y.detach(), eps.grad.clone(), tw.grad.clone(), tb.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    # (round 5: with inputs that need a gradient the wide cases take the ONE-PASS backward -- another weight-gradient kernel than the
    # d-pre-activation route behind gin_dense: agreement to fp32 rounding instead of bit equality)
    close(res[0][2], res[1][2].cpu().numpy(), atol=0, rel=4e-6, what="gin_dense dW: dot route vs separate ops")
    close(res[0][3], res[1][3].cpu().numpy(), atol=0, rel=4e-6, what="gin_dense dbias: dot route vs separate ops")
    # fp64: y and d epsilon
    A = np.zeros((T, N, N))
    for t, chans in enumerate(adjs):
        idx, val, _ = chans[0]
        np.add.at(A[t], (np.asarray(idx)[:, 0], np.asarray(idx)[:, 1]), np.asarray(val, np.float64))
    x64 = x.astype(np.float64)
    agg = 0.3 * x64 + A @ x64
    pre = agg.reshape(T * N, d) @ w.astype(np.float64) + b
    f = {"relu": lambda v: np.maximum(v, 0), "sigmoid": lambda v: 1 / (1 + np.exp(-v))}[act]
    yr = f(pre)
    ygpu = res[0][0].cpu().numpy().reshape(T * N, dout).astype(np.float64)
    dpre = gy.reshape(T * N, dout) * ((ygpu > 0) if act == "relu" else yr * (1 - yr))
    deps = float(((dpre @ w.astype(np.float64).T) * x64.reshape(T * N, d)).sum())
    scale = float(np.abs(((dpre @ w.astype(np.float64).T) * x64.reshape(T * N, d))).sum())
    close(res[0][0], yr.reshape(T, N, dout), atol=2e-6 * max(1.0, float(np.abs(yr).max())), rel=2e-6, what="gin_dense y")
    for r, name in ((res[0], "fused"), (res[1], "two ops")):
        assert abs(float(r[1]) - deps) <= 2e-6 * scale, (name, float(r[1]), deps, scale)


def test_f16_two_piece_rows_spanning_2_to_16_stay_within_fp32_arithmetic():
    """VERDICT r04 item 4.  The f16 x 2 GEMMs (gemmh.hip, >= 16,384 rows) scale every row (forward / dX) or column (weight gradient)
    to its largest magnitude; an element 2^16 below that maximum keeps fewer than 22 bits.  Rows / columns whose magnitudes span
    2^16 (log-uniform inside the row, so both ends are populated): the error against fp64 -- measured in units of the natural
    scale sum_k |x_k| |w_k| of each output -- must stay within TWICE that of numpy's own float32 matmul on the same inputs."""
    from kgcn_amd import ops
    rng = np.random.default_rng(216)
    M, din, dout = 16500, 256, 256
    spread = lambda shape: (2.0 ** rng.uniform(-16, 0, size=shape)).astype(np.float32)
    x = (rng.standard_normal((M, din)) * spread((M, din))).astype(np.float32)
    g = (rng.standard_normal((M, dout)) * spread((M, dout))).astype(np.float32)
    w = (K.glorot_uniform(rng, din, dout) * spread((din, dout))).astype(np.float32)
    for a in (x, g, w):                                           # every row / column really spans the 2^16
        mags = np.abs(a)
        assert (mags.max(axis=-1) / np.maximum(mags.min(axis=-1), 1e-300)).min() >= 2.0 ** 12
    tx, tw = t32(x).requires_grad_(True), t32(w).requires_grad_(True)
    assert ops.lib.kgcn_dense_mfma_products(0, M, din, dout) == 3 and ops.lib.kgcn_dense_mfma_products(2, M, din, dout) == 3
    y = ops.dense(tx, tw, None)
    y.backward(t32(g))
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
    cases = {"fwd": (y.detach().cpu().numpy(), x64 @ w64, x @ w, np.abs(x64) @ np.abs(w64)),
             "dX": (tx.grad.cpu().numpy(), g64 @ w64.T, g @ w.T, np.abs(g64) @ np.abs(w64).T),
             "dW": (tw.grad.cpu().numpy(), x64.T @ g64, x.T @ g, np.abs(x64).T @ np.abs(g64))}
    for name, (got, ref, np32, scale) in cases.items():
        e_hip = float((np.abs(got - ref) / scale).max())
        e_np = float((np.abs(np32.astype(np.float64) - ref) / scale).max())
        import conftest
        factor = 2                                  # (measured, profiles/r05_accuracy.json: 0.67 / 0.68 / 0.58 of numpy's own error)
        conftest.record_accuracy("span 2^16 %s: err / sum|x||w| (tolerance = %d x numpy float32's)" % (name, factor), e_hip, factor * e_np, 1.0)
        assert e_hip <= factor * e_np, (name, e_hip, e_np)
