"""Parity of the HIP path (through the C ABI, via kgcn_amd) against the oracle, on the MI355X.

Tolerance: north_star states 1e-5 (fp32) against the reference CPU path.  Forward activations and
per-batch gradients at cfg1-like sizes are held to max-abs 1e-5 against the fp64 oracle; sums
over >= 10^3 graphs (dW, dbias at benchmark sizes) to 1e-5 * max|ref| (hard parts).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_oracle as K

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def t32(a):
    return torch.as_tensor(np.asarray(a, np.float32), device=dev())


def close(got, ref, atol=ATOL, rel=0.0, what=""):
    """max-abs comparison; every call also records (measured error, tolerance, magnitude of the reference) under the running
    test's id -- conftest.py writes the record to gpurun_out/accuracy_tests.json at the end of the session (the round's copy:
    profiles/r04_accuracy.json), so that a tolerance can be read against what was measured."""
    import conftest
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    mag = float(np.abs(ref).max()) if ref.size else 0.0
    written = atol + rel * mag
    bound = conftest.accuracy_bound(what, mag)           # ten times what this comparison measured (tests/conftest.py: the ratchet)
    tol = written if bound is None else min(written, bound)
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    conftest.record_accuracy(what, err, tol, mag, written)
    assert err <= tol, "%s: max abs err %.3e > %.3e (%s)" % (what, err, tol, "written tolerance" if tol == written else
                                                             "10x the recorded error of this comparison; written tolerance %.3e" % written)
    return err


def synthetic_batch(kind="b30", channels="plain"):
    z = load_golden("g3_synthetic_feed_%s.npz" % kind)
    adjs = unflatten_adjs(z, "adj_")
    x = z["features"]
    if channels != "plain":
        raw = load_golden("g1_synthetic_raw.npz")
        full, _, _ = K.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10},
                                  split_adj_flag=(channels in ("split", "split_norm")),
                                  normalize_adj_flag=(channels in ("norm", "split_norm")))
        adjs = K.feed_batch(list(z["batch_idx"]), 30, full)["adjs"]
    return x, adjs


def random_graphs(rng, T, N, density=0.12, empty_every=0, dup=False):
    adjs = []
    for t in range(T):
        if empty_every and t % empty_every == empty_every - 1:
            adjs.append([(np.zeros((0, 2), np.int32), np.zeros((0,), np.float32), [N, N])])
            continue
        a = (rng.random((N, N)) < density) * rng.standard_normal((N, N))
        a[rng.integers(0, N)] = 0                      # an all-zero row
        idx, val, shp = K.dense_to_sparse(a)
        idx = np.asarray(idx).reshape(-1, 2)
        if dup and len(val):
            idx = np.concatenate([idx, idx[:3]])       # duplicated entries accumulate
            val = np.concatenate([val, val[:3]])
            p = rng.permutation(len(val))              # and unsorted COO order
            idx, val = idx[p], val[p]
        adjs.append([(idx.astype(np.int32), val.astype(np.float32), [N, N])])
    return adjs


# ---------------------------------------------------------------------------------------------
# Bspmm / Bspmdt / Bconv / values gradient
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 3, 50, 64, 128, 200, 256])
def test_bspmm_synthetic_jbl(D):
    from kgcn_amd import BatchedCSR, ops
    x, adjs = synthetic_batch()
    rng = np.random.default_rng(D)
    rhs = rng.standard_normal((30, 10, D)).astype(np.float32)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], device=dev())
    assert (csr.num_graphs, csr.rows, csr.cols) == (30, 10, 10)
    out = ops.bspmm(csr, t32(rhs))
    ref = np.stack(K.bspmm([a[0] for a in adjs], list(rhs)))
    close(out, ref, what="bspmm")
    assert torch.all(out[10:] == 0)                    # dummy graphs -> zeros
    outT = ops.bspmm(csr.transpose(), t32(rhs))
    refT = np.stack(K.bspmm([a[0] for a in adjs], list(rhs), adjoint_a=True))
    close(outT, refT, what="bspmm^T")


@pytest.mark.parametrize("N,D,dup", [(32, 64, False), (32, 64, True), (50, 128, False), (7, 12, True),
                                     (64, 32, False), (200, 64, False), (33, 8, True)])
def test_bspmm_random(N, D, dup):
    from kgcn_amd import BatchedCSR, ops
    rng = np.random.default_rng(N * 1000 + D)
    T = 37
    adjs = random_graphs(rng, T, N, empty_every=5, dup=dup)
    rhs = rng.standard_normal((T, N, D)).astype(np.float32)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    ref = np.stack(K.bspmm([a[0] for a in adjs], list(rhs)))
    close(ops.bspmm(csr, t32(rhs)), ref, rel=2e-6, what="bspmm")
    refT = np.stack(K.bspmm([a[0] for a in adjs], list(rhs), adjoint_a=True))
    close(ops.bspmm(csr.transpose(), t32(rhs)), refT, rel=2e-6, what="bspmm^T")


def test_bspmm_rectangular_and_block_diagonal():
    """M != K, and the config-3 shape: ONE [sumN x sumN] block-diagonal matrix with batch 1
    (kgcn/data_util.py:698-845, example_model/sparse.py:65-69)."""
    from kgcn_amd import BatchedCSR, ops
    rng = np.random.default_rng(5)
    a = (rng.random((9, 13)) < 0.3) * rng.standard_normal((9, 13))
    idx, val, _ = K.dense_to_sparse(a)
    csr = BatchedCSR.from_coo_list([(idx, val, [9, 13])] * 3, device=dev())
    rhs = rng.standard_normal((3, 13, 20)).astype(np.float32)
    close(ops.bspmm(csr, t32(rhs)), np.stack([a @ rhs[i] for i in range(3)]), what="rect")
    # block diagonal: 128 graphs x 50 nodes, D = 128
    adjs = K.synth_mol_graphs(rng, 128, 50, 5, normalize=True)
    big = K.block_diag_csr(adjs, 0, 50).tocoo()
    csr = BatchedCSR.from_arrays(np.zeros(big.nnz, np.int64), big.row, big.col, big.data, 1, 6400, 6400,
                                 device=dev())
    x = rng.standard_normal((6400, 128)).astype(np.float32)
    close(ops.bspmm(csr, t32(x)), big.tocsr() @ x.astype(np.float64), what="blockdiag")


def test_bspmm_autograd_and_values_grad():
    from kgcn_amd import BatchedCSR, ops
    x, adjs = synthetic_batch("full30", "norm")
    rng = np.random.default_rng(7)
    al = [a[0] for a in adjs]
    rhs = rng.standard_normal((30, 10, 24)).astype(np.float32)
    g = rng.standard_normal((30, 10, 24)).astype(np.float32)
    csr = BatchedCSR.from_coo_list(al, device=dev())
    r = t32(rhs).requires_grad_(True)
    v = csr.values.clone().requires_grad_(True)
    out = ops.bspmm(csr, r, v)
    out.backward(t32(g))
    vg, rg = K.bspmm_grad(al, list(rhs), list(g))
    close(r.grad, np.stack(rg), what="d rhs")
    close(v.grad, np.concatenate(vg), what="d values")      # COO already row-major -> same order


def test_op_wrappers_api():
    """Reference call conventions: kgcn/bspmm_call.py:11-16, bconv_call.py:11-23,
    batched_call.py:18-27 (lists in, lists out)."""
    from kgcn_amd.bspmm_call import BatchedSpMM
    from kgcn_amd.bconv_call import BatchedConv
    from kgcn_amd.batched_call import BatchedSpMDT
    import collections
    SparseTensorValue = collections.namedtuple("SparseTensorValue", ["indices", "values", "dense_shape"])
    x, adjs = synthetic_batch("b30", "split")
    rng = np.random.default_rng(8)
    B, C, D = 30, 6, 16
    sp = [[SparseTensorValue(*a) for a in row] for row in adjs]
    dense = [[rng.standard_normal((10, D)).astype(np.float32) for _ in range(C)] for _ in range(B)]
    td = [[t32(d) for d in row] for row in dense]
    o = BatchedSpMM().call([sp[b][2] for b in range(B)], [td[b][2] for b in range(B)])
    assert isinstance(o, list) and len(o) == B and tuple(o[0].shape) == (10, D)
    close(torch.stack(o), np.stack(K.bspmm([adjs[b][2] for b in range(B)], [dense[b][2] for b in range(B)])))
    o = BatchedSpMM().call([sp[b][2] for b in range(B)], [td[b][2].t().contiguous() for b in range(B)],
                           adjoint_a=True, adjoint_b=True)
    close(torch.stack(o), np.stack(K.bspmm([adjs[b][2] for b in range(B)], [dense[b][2] for b in range(B)],
                                           adjoint_a=True)))
    o = BatchedConv().call(sp, td)
    assert len(o) == B
    close(torch.stack(o), np.stack(K.bconv(adjs, dense)), what="bconv")
    stacked = np.concatenate([dense[b][3] for b in range(B)], 0)
    o = BatchedSpMDT().call([sp[b][3] for b in range(B)], t32(stacked))
    assert len(o) == B
    close(torch.stack(o), np.stack(K.bspmdt([adjs[b][3] for b in range(B)], stacked)), what="bspmdt")
    # differentiable .values through the wrapper (visualization's use, bspmm_call.py:50-55)
    vals = [t32(adjs[b][3][1]).requires_grad_(True) for b in range(B)]
    sp3 = [SparseTensorValue(adjs[b][3][0], vals[b], adjs[b][3][2]) for b in range(B)]
    r = [t32(dense[b][3]).requires_grad_(True) for b in range(B)]
    o = BatchedSpMM().call(sp3, r)
    g = rng.standard_normal((B, 10, D)).astype(np.float32)
    (torch.stack(o) * t32(g)).sum().backward()
    vg, rg = K.bspmm_grad([adjs[b][3] for b in range(B)], [dense[b][3] for b in range(B)], list(g))
    for b in range(B):
        close(vals[b].grad, vg[b], what="d values[%d]" % b)
        close(r[b].grad, rg[b], what="d rhs[%d]" % b)


def _np_act(v, name):
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))
    if name == "relu":
        return np.maximum(v, 0)
    if name == "tanh":
        return np.tanh(v)
    return v


def _np_dact(a, name):
    return {"sigmoid": a * (1 - a), "relu": (a > 0).astype(np.float64), "tanh": 1 - a * a}.get(name, np.ones_like(a))


@pytest.mark.parametrize("act", ["sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("N,D,channels", [(10, 16, "split"), (10, 50, "split"), (50, 256, "one"), (10, 64, "norm")])
def test_fused_activation_aggregation(act, N, D, channels):
    """out = act(sum_c A_c @ rhs_c) in ONE launch (channel loop + activation epilogue; tile kernel for small graphs with
    d % 4 == 0, gather kernel otherwise) and its backward (act' formed while the gradient is gathered), against the
    oracle: bconv followed by the model's elementwise op (example_model/model.py:42-43)."""
    from kgcn_amd import BatchedAdjacency, ops
    rng = np.random.default_rng(N * D)
    if channels == "one":
        B = 40
        adjs = K.synth_mol_graphs(rng, B, N, 3, normalize=True)
    else:
        _, adjs = synthetic_batch("b30", channels)
        B = 30
    C = len(adjs[0])
    rhs = [[rng.standard_normal((N, D)).astype(np.float32) for _ in range(C)] for _ in range(B)]
    g = rng.standard_normal((B, N, D)).astype(np.float32)
    adj = BatchedAdjacency.from_adjs(adjs, n_nodes=N, device=dev())
    cat = np.stack([np.concatenate(r, 1) for r in rhs]).reshape(B * N, C * D)
    tr = t32(cat).requires_grad_(True)
    out = ops.bconv(adj, tr, D, activation=act)
    pre = np.stack(K.bconv(adjs, rhs))
    ref = _np_act(pre, act)
    close(out.reshape(B, N, D), ref, rel=1e-6, what="act(bconv) " + act)
    out.backward(t32(g).reshape(B * N, D))
    # relu': read from the kernel's own output, like tf.nn.relu's gradient (a pre-activation within one rounding of 0 may
    # land on either side in fp32)
    aout = out.detach().cpu().numpy().reshape(np.shape(ref)).astype(np.float64) if act == "relu" else ref
    _, rg = K.bconv_grad(adjs, rhs, list(g * _np_dact(aout, act)))
    want = np.stack([np.concatenate(r, 1) for r in rg]).reshape(B * N, C * D)
    close(tr.grad, want, rel=1e-6, what="d rhs through act(bconv) " + act)


@pytest.mark.parametrize("act", ["sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("M,din,dout", [(300, 50, 50), (1000, 64, 64), (777, 81, 256), (513, 256, 256), (64, 512, 96),
                                        (3001, 256, 256), (2050, 320, 200), (1500, 256, 512), (2000, 256, 50), (1100, 160, 64),
                                        (1500, 132, 7)])
def test_fused_activation_dense(act, M, din, dout):
    """y = act(x W + b) in the GEMM epilogue of all three dense kernels (f32-MFMA tiled / persistent, bf16-split), and
    the backward through it, against numpy."""
    from kgcn_amd import ops
    rng = np.random.default_rng(M + din)
    x = rng.standard_normal((M, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = rng.standard_normal(dout).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    y = ops.dense(tx, tw, tb, activation=act)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    ref = _np_act(x64 @ w64 + b, act)
    close(y, ref, rel=2e-6, what="act(dense) " + act)
    y.backward(t32(g))
    gz = g * _np_dact(y.detach().cpu().numpy().astype(np.float64) if act == "relu" else ref, act)
    close(tx.grad, gz @ w64.T, rel=2e-6, what="dx")
    close(tw.grad, x64.T @ gz, rel=1e-5, what="dw")
    close(tb.grad, gz.sum(0), rel=1e-5, what="db")
    # the stand-alone activation op
    tz = t32(x).requires_grad_(True)
    a = ops.activation(tz, act)
    close(a, _np_act(x64, act), rel=1e-6)
    a.backward(t32(x))
    close(tz.grad, x64 * _np_dact(_np_act(x64, act), act), rel=1e-6)


def test_dense_entry_points_with_padded_leading_dimensions():
    """The C ABI takes leading dimensions: the wide-input / narrow-output kernels (gemmn.hip, wgradn.hip) and the wide GEMMs
    on operands that are column slices of wider tensors (x_ld > din, y_ld > dout, dy_ld > dout)."""
    from kgcn_amd._lib import lib, check, ptr, current_stream
    rng = np.random.default_rng(5)
    for M, din, dout in ((5000, 256, 50), (4200, 160, 64), (4300, 256, 256)):
        xl, yl = din + 8, dout + 12
        xbig = t32(rng.standard_normal((M, xl)).astype(np.float32))
        gbig = t32(rng.standard_normal((M, yl)).astype(np.float32))
        w = t32(K.glorot_uniform(rng, din, dout)); b = t32(rng.standard_normal(dout).astype(np.float32))
        ybig = torch.full((M, yl), 7.0, device=dev())
        wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
        ws = torch.empty(max(wsb, 4) // 4, device=dev())
        check(lib.kgcn_dense_fwd_ws_f32(ptr(xbig), M, din, xl, ptr(w), dout, 0, ptr(b), ptr(ybig), dout, yl, 0, ptr(ws), wsb,
                                        current_stream()))
        x64, w64 = xbig[:, :din].double().cpu().numpy(), w.double().cpu().numpy()
        close(ybig[:, :dout], x64 @ w64 + b.double().cpu().numpy(), rel=2e-6, what="y (ld)")
        assert bool((ybig[:, dout:] == 7.0).all()), "columns beyond dout were written"
        dw = torch.empty(din, dout, device=dev()); db = torch.empty(dout, device=dev())
        wb = lib.kgcn_dense_wgrad_workspace_bytes(M, din, dout)
        wsw = torch.empty(max(wb, 4) // 4, device=dev())
        check(lib.kgcn_dense_wgrad_f32(ptr(xbig), xl, ptr(gbig), yl, M, din, dout, ptr(dw), ptr(db), ptr(wsw), wb,
                                       current_stream()))
        g64 = gbig[:, :dout].double().cpu().numpy()
        close(dw, x64.T @ g64, rel=1e-5, what="dw (ld)")
        close(db, g64.sum(0), rel=1e-5, what="db (ld)")


def test_bconv_and_bspmdt_values_gradients():
    """kgcn/bconv_call.py:55-67 and kgcn/batched_call.py:66-73 register a gradient for the sparse VALUES of every
    graph-channel (gather rows of the output gradient, gather rows of the dense operand, multiply, reduce): through
    BatchedConv().call / BatchedSpMDT().call with differentiable .values, against the oracle's restatement.
    (The reference's Bspmdt version indexes ROWS of the stacked operand where it means per-graph blocks -- b[i] for
    i < numTensors, batched_call.py:58 -- and would not broadcast; the mathematically meant gradient is taken.)"""
    from kgcn_amd.bconv_call import BatchedConv
    from kgcn_amd.batched_call import BatchedSpMDT
    import collections
    STV = collections.namedtuple("SparseTensorValue", ["indices", "values", "dense_shape"])
    x, adjs = synthetic_batch("b30", "split")               # 6 channels, 10 real + 20 dummy graphs
    rng = np.random.default_rng(21)
    B, C, D = 30, 6, 12
    dense = [[rng.standard_normal((10, D)).astype(np.float32) for _ in range(C)] for _ in range(B)]
    g = rng.standard_normal((B, 10, D)).astype(np.float32)
    vals = [[t32(adjs[b][c][1]).requires_grad_(c != 4) for c in range(C)] for b in range(B)]   # channel 4: constants
    sp = [[STV(adjs[b][c][0], vals[b][c], adjs[b][c][2]) for c in range(C)] for b in range(B)]
    td = [[t32(d).requires_grad_(True) for d in row] for row in dense]
    out = BatchedConv().call(sp, td)
    close(torch.stack(out), np.stack(K.bconv(adjs, dense)), what="bconv fwd")
    (torch.stack(out) * t32(g)).sum().backward()
    vg, rg = K.bconv_grad(adjs, dense, list(g))
    for b in range(B):
        for c in range(C):
            close(td[b][c].grad, rg[b][c], what="bconv d rhs[%d][%d]" % (b, c))
            if c == 4:
                assert vals[b][c].grad is None
            else:
                close(vals[b][c].grad, vg[b][c], what="bconv d values[%d][%d]" % (b, c))
    # Bspmdt: one stacked dense operand, list of outputs
    stacked = np.concatenate([dense[b][3] for b in range(B)], 0)
    v3 = [t32(adjs[b][3][1]).requires_grad_(True) for b in range(B)]
    ts = t32(stacked).requires_grad_(True)
    o = BatchedSpMDT().call([STV(adjs[b][3][0], v3[b], adjs[b][3][2]) for b in range(B)], ts)
    (torch.stack(o) * t32(g)).sum().backward()
    vg3, rg3 = K.bspmdt_grad([adjs[b][3] for b in range(B)], stacked, list(g))
    close(ts.grad, rg3, what="bspmdt d rhs")
    for b in range(B):
        close(v3[b].grad, vg3[b], what="bspmdt d values[%d]" % b)


# ---------------------------------------------------------------------------------------------
# dense contraction (GraphDense)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,din,dout", [(300, 3, 50), (300, 50, 50), (1000, 64, 64), (777, 81, 256),
                                        (129, 256, 50), (5, 7, 2), (4096, 128, 128), (10000, 256, 256),
                                        (333, 100, 300), (2049, 259, 130), (64, 512, 96), (5003, 256, 50), (4100, 128, 64),
                                        (4500, 200, 7), (20000, 512, 2), (4096, 50, 12), (128, 256, 10), (3000, 1024, 5), (777, 64, 16),
                                        (4096, 12, 50), (1000, 3, 64), (513, 16, 300),
                                        # launch shapes of the table GEMM (gemm3.hip: g3_table_launch): 64- / 128-column blocks over
                                        # few rows, with a ragged last block; whole rounds + a narrow-block launch over the rest
                                        (1500, 256, 200), (12000, 300, 256), (36160, 256, 256), (47000, 256, 200),
                                        # narrow input, wide output: the register-split weight gradient (wgradx.hip), ragged last 16-row step
                                        (50001, 81, 256), (4099, 96, 130)])
def test_dense_fwd_bwd(M, din, dout):
    from kgcn_amd import ops
    rng = np.random.default_rng(M + din)
    x = rng.standard_normal((M, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = rng.standard_normal((dout,)).astype(np.float32)
    g = rng.standard_normal((M, dout)).astype(np.float32)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    y = ops.dense(tx, tw, tb)
    y.backward(t32(g))
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), g.astype(np.float64)
    close(y, x64 @ w64 + b, rel=1e-6, what="dense fwd")
    close(tx.grad, g64 @ w64.T, rel=1e-6, what="dense dx")
    close(tw.grad, x64.T @ g64, rel=2e-6, what="dense dw")
    close(tb.grad, g64.sum(0), rel=2e-6, what="dense db")


# ---------------------------------------------------------------------------------------------
# GraphConv layer: synthetic.jbl (cfg1), all dispatch variants, model.py layer stack
# ---------------------------------------------------------------------------------------------
def _set_variant(name):
    from kgcn_amd import layers
    import types
    layers.load_bspmm(types.SimpleNamespace(batched=name == "batched", bspmm=name == "bspmm",
                                            bconv=name == "bconv"))


@pytest.mark.parametrize("variant", ["default", "bspmm", "bconv", "batched"])
@pytest.mark.parametrize("channels", ["plain", "norm", "split"])
def test_graphconv_layer_synthetic_jbl(variant, channels):
    from kgcn_amd import layers
    x, adjs = synthetic_batch("b30", channels)
    C = len(adjs[0])
    rng = np.random.default_rng(11)
    try:
        _set_variant(variant)
        layer = layers.GraphConv(50, C)
        tx = t32(x).requires_grad_(True)
        out = layer(tx, adj=adjs)
        assert tuple(out.shape) == (30, 10, 50) == tuple(layer.compute_output_shape((30, 10, 3)))
        w = [p.detach().cpu().numpy() for p in layer.w]
        assert all(tuple(p.shape) == (3, 50) for p in layer.w) and all(tuple(p.shape) == (1, 50) for p in layer.bias)
        lim = np.sqrt(6.0 / 53)
        assert all(np.abs(t).max() <= lim for t in w) and all(float(p.abs().max()) == 0 for p in layer.bias)
        with torch.no_grad():
            for p in layer.bias:
                p.copy_(t32(rng.standard_normal((1, 50))))
        b = [p.detach().cpu().numpy() for p in layer.bias]
        out = layer(tx, adj=adjs)
        ref = K.graphconv_fwd(x, adjs, w, b)
        close(out, ref, what="graphconv fwd")
        g = rng.standard_normal(ref.shape).astype(np.float32)
        out.backward(t32(g))
        dx, dw, db = K.graphconv_bwd(x, adjs, w, b, g)
        close(tx.grad, dx, what="dX")
        for c in range(C):
            close(layer.w[c].grad, dw[c], what="dW%d" % c)
            close(layer.bias[c].grad, db[c], what="dbias%d" % c)
    finally:
        _set_variant("default")


def test_model_py_layer_stack():
    """example_model/model.py:41-46: GraphConv(50) -> sigmoid -> GraphConv(50) -> sigmoid ->
    GraphConv(50), checked layer by layer, then GraphDense(50) -> sigmoid -> GraphGather."""
    from kgcn_amd import layers
    x, adjs = synthetic_batch("full30", "plain")
    convs = [layers.GraphConv(50, 1) for _ in range(3)]
    dense = layers.GraphDense(50)
    h = t32(x)
    ref = x.astype(np.float64)
    sig = lambda a: 1.0 / (1.0 + np.exp(-a))
    for i, conv in enumerate(convs):
        h = conv(h, adj=adjs)
        w = [conv.w[0].detach().cpu().numpy()]
        b = [conv.bias[0].detach().cpu().numpy()]
        ref = K.graphconv_fwd(ref, adjs, w, b)
        close(h, ref, what="layer %d" % i)
        if i < 2:
            h = torch.sigmoid(h)
            ref = sig(ref)
    h = torch.sigmoid(h)
    ref = sig(ref)
    h = dense(h)
    ref = K.graphdense_fwd(ref, dense.kernel.detach().cpu().numpy(), dense.bias.detach().cpu().numpy())
    close(h, ref, what="graphdense")
    out = layers.GraphGather()(torch.sigmoid(h))
    close(out, K.gather_fwd(sig(ref)), what="gather")
    out.sum().backward()
    assert convs[0].w[0].grad is not None and torch.isfinite(convs[0].w[0].grad).all()


# ---------------------------------------------------------------------------------------------
# fused GraphConv kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,din,dout,T", [(32, 64, 64, 300), (32, 64, 64, 5), (10, 4, 52, 30),
                                          (20, 12, 36, 77), (32, 32, 64, 129), (17, 64, 8, 64),
                                          (32, 64, 64, 3001), (10, 3, 50, 30), (10, 50, 50, 200),
                                          (7, 5, 6, 65), (32, 63, 1, 40), (31, 2, 61, 33),
                                          # the two-waves-per-graph backward (graphconv_bwd_pairs_kernel) takes FULL-shape batches of
                                          # >= 2,048 graphs: every pair owns exactly two / one pair owns three / three or four / 2,047: the
                                          # one-wave kernel still
                                          (32, 64, 64, 2048), (32, 64, 64, 2049), (32, 64, 64, 4000), (32, 64, 64, 2047)])
def test_graphconv_fused(N, din, dout, T):
    from kgcn_amd import BatchedCSR, ops
    rng = np.random.default_rng(N + din + dout + T)
    if N == 32:
        adjs = K.synth_mol_graphs(rng, T, 32, 3, normalize=(T % 2 == 1))
        if T > 4:
            adjs[3] = [(np.zeros((0, 2), np.int32), np.zeros((0,), np.float32), [32, 32])]
    else:
        adjs = random_graphs(rng, T, N, density=0.2, empty_every=7, dup=True)
    x = rng.standard_normal((T, N, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout)
    b = rng.standard_normal((1, dout)).astype(np.float32)
    g = rng.standard_normal((T, N, dout)).astype(np.float32)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    assert ops.graphconv_fused_supported(csr, din, dout)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    out = ops.graphconv_fused(tx, tw, tb, csr)
    ref = K.graphconv_fwd_fast(x, adjs, [w], [b])
    close(out, ref, rel=1e-6, what="fused fwd")
    out.backward(t32(g))
    dx, dw, db = K.graphconv_bwd_fast(x, adjs, [w], g)
    close(tx.grad, dx, rel=1e-6, what="fused dX")
    close(tw.grad, dw[0], rel=1e-5, what="fused dW")
    close(tb.grad, db[0], rel=1e-5, what="fused dbias")
    # the fused kernels and the unfused kernels are two HIP implementations of the same function
    fw = ops.dense(t32(x).reshape(T * N, din), t32(w), t32(b))
    out2 = ops.bspmm(csr, fw).reshape(T, N, dout)
    close(out2, ref, rel=1e-6, what="unfused fwd")


def test_pairs_backward_repeats_bit_for_bit():
    """graphconv_bwd_pairs_kernel hands LDS data between the two waves of a pair (planes, gather tile) under one workgroup barrier per
    graph and one flag; a missed ordering would show as run-to-run differences long before it shows against a tolerance.  300 launches on
    the same operands (5,000 graphs: pairs of 4 and 5 graphs), every result bit-equal to the first; dX and dW also bit-equal to the
    one-wave-per-graph kernel's (same products in the same order), which a 2,047-graph prefix of the batch still takes."""
    from kgcn_amd import BatchedCSR
    from kgcn_amd._lib import lib, ptr, current_stream, check
    rng = np.random.default_rng(99)
    T, N, D = 5000, 32, 64
    adjs = K.synth_mol_graphs(rng, T, N, 3, normalize=True)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    at = csr.transpose().padded4()
    x = t32(rng.standard_normal((T, N, D)).astype(np.float32))
    g = t32(rng.standard_normal((T, N, D)).astype(np.float32))
    w = t32(K.glorot_uniform(rng, D, D))
    wsb = lib.kgcn_graphconv_bwd_workspace_bytes(T, D, D)
    wsp = torch.empty(wsb // 4, device=dev())

    def run(desc, n):
        dx = torch.full((n, N, D), float("nan"), device=dev())
        dw = torch.empty((D, D), device=dev()); db = torch.empty(D, device=dev())
        check(lib.kgcn_graphconv_bwd_f32(desc, ptr(x), ptr(w), ptr(g), D, D, ptr(dx), ptr(dw), ptr(db), ptr(wsp), wsb, current_stream()),
              "kgcn_graphconv_bwd_f32")
        return dx, dw, db

    first = run(at.desc(), T)
    for rep in range(300):
        again = run(at.desc(), T)
        for a, b, name in zip(first, again, ("dX", "dW", "dbias")):
            assert torch.equal(a, b), "launch %d: %s differs from the first launch" % (rep, name)
    sub = BatchedCSR.from_coo_list([a[0] for a in adjs[:2047]], rows=N, cols=N, device=dev()).transpose().padded4()
    dx_p, _, _ = run(sub.desc(), 2047)                       # the planes kernel on a prefix: dX rows are per-graph quantities
    assert torch.equal(dx_p, first[0][:2047])


def _row_close(got, ref, rel, what):
    """every row within rel * (largest magnitude of that reference row): inputs spanning 60 decades make one global
    maximum meaningless"""
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    bad = np.abs(got - ref) > rel * scale + 1e-37
    assert not bad.any(), "%s: %d elements beyond %.1e of their row maximum (worst %.3e)" % (
        what, int(bad.sum()), rel, float((np.abs(got - ref) / (scale + 1e-300)).max()))


@pytest.mark.parametrize("T", [130, 2200])
@pytest.mark.parametrize("case", ["exponents", "full_significands", "denormals"])
def test_bf16_split_edge_values(case, T):
    """The fused FULL-shape kernels contract on the bf16 matrix pipe with an exact 3-way split of every fp32 operand
    (kgcn_common.h split_pair) and claim fp32 accuracy.  Stress the claim where a split could lose bits: per-row
    exponents from 1e-18 to 1e18 (dW multiplies them by gradients of 1e-9 .. 1e9: 1e27 stays finite), operands with all 24 significand bits set (every piece saturated, every cross
    product non-zero), and fp32 denormals (their low pieces are bf16 denormals).  Reference: fp64 oracle; second HIP
    implementation: the unfused f32-MFMA dense kernel + Bspmm (kgcn_dense_fwd_f32, exact fp32 products)."""
    from kgcn_amd import BatchedCSR, ops
    rng = np.random.default_rng({"exponents": 1, "full_significands": 2, "denormals": 3}[case])
    N, D = 32, 64                      # T = 2,200: the backward runs as graphconv_bwd_pairs_kernel (two waves per graph)
    adjs = K.synth_mol_graphs(rng, T, N, 3, normalize=True)
    x = rng.standard_normal((T, N, D)).astype(np.float32)
    g = rng.standard_normal((T, N, D)).astype(np.float32)
    w = K.glorot_uniform(rng, D, D)
    if case == "exponents":
        # one scale per node row of x (all inside one graph) and per graph of g (dW sums products of both); within a row all 64 values share the scale, so every output row has a well defined size
        x *= (10.0 ** rng.uniform(-18, 18, size=(T, N, 1))).astype(np.float32)
        g *= (10.0 ** rng.uniform(-9, 9, size=(T, 1, 1))).astype(np.float32)
    elif case == "full_significands":
        ones = lambda a: (a.view(np.uint32) | np.uint32(0x007fffff)).view(np.float32)
        x, g, w = ones(x), ones(g), ones(w)
    else:
        tiny = np.float32(2.0 ** -140)                         # fp32 denormals (the smallest normal is 2^-126)
        x[:, ::2, :] = (x[:, ::2, :] * tiny).astype(np.float32)
        assert (np.abs(x[:, ::2, :]) < 2.0 ** -126).all() and (x[:, ::2, :] != 0).any()
    b = np.zeros((1, D), np.float32)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    out = ops.graphconv_fused(tx, tw, tb, csr)
    out.backward(t32(g))
    ref = K.graphconv_fwd_fast(x, adjs, [w], [b])
    dx, dw, db = K.graphconv_bwd_fast(x, adjs, [w], g)
    assert np.isfinite(ref).all() and np.isfinite(dx).all() and np.isfinite(dw[0]).all()
    _row_close(out, ref, 2e-6, case + ": fused fwd")
    _row_close(tx.grad, dx, 2e-6, case + ": fused dX")
    if case != "exponents":          # dW / dbias mix rows of every scale: only meaningful against their own maximum
        close(tw.grad, dw[0], atol=0, rel=2e-6, what=case + ": fused dW")
        close(tb.grad, db[0], atol=0, rel=2e-6, what=case + ": fused dbias")
    else:
        _row_close(tw.grad, dw[0], 1e-5, case + ": fused dW")
    ux = t32(x).requires_grad_(True)
    out_u = ops.bspmm(csr, ops.dense(ux.reshape(T * N, D), t32(w), t32(b))).reshape(T, N, D)
    _row_close(out_u, ref, 2e-6, case + ": unfused fwd")
    # the two HIP implementations agree to a few fp32 roundings of the row maximum
    _row_close(out, out_u.detach().cpu().numpy(), 3e-6, case + ": fused vs unfused")


@pytest.mark.parametrize("T", [66, 2100])
def test_bf16_split_non_finite_inputs(T):
    """Documented behaviour (include/kgcn_hip.h): the split of +-inf is (inf, nan, nan), so a non-finite
    input value makes every output element that depends on it NaN or +-inf (the f32 kernels and TF give +-inf or NaN
    there), and leaves every other element untouched -- never a silently finite wrong value.  (2,100 graphs: the backward
    with two waves per graph slot, graphconv_bwd_pairs_kernel.)"""
    from kgcn_amd import BatchedCSR, ops
    rng = np.random.default_rng(11)
    N, D = 32, 64
    adjs = K.synth_mol_graphs(rng, T, N, 3)
    x = rng.standard_normal((T, N, D)).astype(np.float32)
    w = K.glorot_uniform(rng, D, D)
    b = np.zeros((1, D), np.float32)
    g = rng.standard_normal((T, N, D)).astype(np.float32)
    bad = [(5, 7, 3, np.inf), (5, 9, 60, -np.inf), (40, 0, 0, np.nan)]
    xb = x.copy()
    for t, n, k, v in bad:
        xb[t, n, k] = v
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    tx, tw, tb = t32(xb).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    out = ops.graphconv_fused(tx, tw, tb, csr)
    out.backward(t32(g))
    ref = K.graphconv_fwd_fast(x, adjs, [w], [b])                  # clean reference
    touched = np.zeros((T, N), bool)                               # output rows that aggregate a poisoned node
    for t, n, _, _ in bad:
        idx = np.asarray(adjs[t][0][0]).reshape(-1, 2)
        touched[t, idx[idx[:, 1] == n, 0]] = True
    o = out.detach().cpu().numpy()
    assert not np.isfinite(o[touched]).any(), "an output row that aggregates a non-finite node must be non-finite"
    np.testing.assert_allclose(o[~touched], ref[~touched], rtol=0, atol=1e-5)
    # backward: dX = (A^T g) W^T does not depend on x at all; dW = sum x^T dFW is non-finite in the poisoned rows of x
    dx, _, _ = K.graphconv_bwd_fast(x, adjs, [w], g)
    np.testing.assert_allclose(tx.grad.cpu().numpy(), dx, rtol=0, atol=1e-5)
    dwn = tw.grad.cpu().numpy()
    rows = sorted({k for _, _, k, _ in bad})
    assert not np.isfinite(dwn[rows]).any()
    assert np.isfinite(np.delete(dwn, rows, axis=0)).all()


def test_graphconv_layer_uses_fused_and_matches():
    from kgcn_amd import layers
    rng = np.random.default_rng(3)
    T = 64
    adjs = K.synth_mol_graphs(rng, T, 32, 3)
    x = rng.standard_normal((T, 32, 64)).astype(np.float32)
    layer = layers.GraphConv(64, 1)
    tx = t32(x).requires_grad_(True)
    out = layer(tx, adjs)                                 # positional adj (example_model/sparse.py:69)
    w = [layer.w[0].detach().cpu().numpy()]
    b = [layer.bias[0].detach().cpu().numpy()]
    close(out, K.graphconv_fwd(x, adjs, w, b), rel=1e-6)
    assert out.grad_fn.__class__.__name__.startswith("_GraphConvFused")


# ---------------------------------------------------------------------------------------------
# GINAggregate, GraphGather, GraphDense ragged
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("channels,D", [("plain", 3), ("split", 64), ("norm", 256)])
def test_gin_aggregate(channels, D):
    from kgcn_amd import layers
    _, adjs = synthetic_batch("b30", channels)
    C = len(adjs[0])
    rng = np.random.default_rng(D)
    x = rng.standard_normal((30, 10, D)).astype(np.float32)
    layer = layers.GINAggregate(C)
    tx = t32(x).requires_grad_(True)
    out = layer(tx, adj=adjs)
    assert all(float(e) == 0 for e in layer.epsilon)
    close(out, K.gin_fwd(x, adjs, [0.0] * C), what="gin eps=0")
    eps = rng.standard_normal(C).astype(np.float32)
    with torch.no_grad():
        for e, v in zip(layer.epsilon, eps):
            e.fill_(float(v))
    out = layer(tx, adj=adjs)
    close(out, K.gin_fwd(x, adjs, eps), what="gin fwd")
    g = rng.standard_normal(x.shape).astype(np.float32)
    out.backward(t32(g))
    dx, deps = K.gin_bwd(x, adjs, eps, g)
    close(tx.grad, dx, what="gin dx")
    # d eps = <g, x>: one fp32 dot product over B*N*D terms -- tolerance = 8 sigma of fp32 summation noise for that length
    terms = g.astype(np.float64) * x
    noise = 8 * 6e-8 * np.sqrt(terms.size) * np.sqrt((terms ** 2).mean())
    for c in range(C):
        close(layer.epsilon[c].grad, deps[c], rel=1e-6, atol=max(2e-5, noise), what="gin deps")
    try:                                                   # quirk Q1: accelerated branches drop eps
        _set_variant("bspmm")
        close(layer(t32(x), adj=adjs), K.gin_fwd(x, adjs, eps, with_eps=False), what="gin no-eps")
    finally:
        _set_variant("default")


def test_graph_gather_and_ragged_dense():
    from kgcn_amd import layers
    rng = np.random.default_rng(9)
    for shape in [(30, 10, 50), (7, 32, 64), (3, 5, 1)]:
        x = rng.standard_normal(shape).astype(np.float32)
        tx = t32(x).requires_grad_(True)
        out = layers.GraphGather()(tx)
        close(out, K.gather_fwd(x), what="gather")
        g = rng.standard_normal(out.shape).astype(np.float32)
        out.backward(t32(g))
        close(tx.grad, K.gather_bwd(g, shape[1]), what="gather bwd")
    x = rng.standard_normal((6, 10, 8)).astype(np.float32)
    en = np.array([10, 3, 0, 7, 1, 10])
    d = layers.GraphDense(5)
    tx = t32(x).requires_grad_(True)
    y = d(tx, enabled_node_nums=en)
    with torch.no_grad():
        d.bias.copy_(t32(rng.standard_normal(5)))
    y = d(tx, enabled_node_nums=en)
    kern, bias = d.kernel.detach().cpu().numpy(), d.bias.detach().cpu().numpy()
    close(y, K.graphdense_ragged_fwd(x, kern, bias, en))
    # backward of the ragged path (kgcn/layers.py:243-254: Dense over the valid rows only, zero padding after the
    # split): gradients flow through the valid rows only -- the oracle's dense backward on the masked output gradient
    gy = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(t32(gy))
    mask = (np.arange(10)[None, :] < en[:, None])[:, :, None]
    dx, dk, db = K.graphdense_bwd(x, kern, gy * mask)
    close(tx.grad, dx, what="ragged GraphDense dX")
    close(d.kernel.grad, dk, rel=1e-6, what="ragged GraphDense dkernel")
    close(d.bias.grad, db, rel=1e-6, what="ragged GraphDense dbias")


@pytest.mark.parametrize("D,ragged", [(50, True), (64, True), (3, False), (256, True), (300, False)])
def test_graph_batch_normalization_both_phases(D, ragged):
    """kgcn/layers.py:170-220 in both Keras learning phases (quirk Q6), ragged enabled_node_nums incl. an empty graph:
    forward, the moving-statistics update, and the backward (through the batch statistics in phase 1) against the
    literal gather / normalise / split / pad restatement of the oracle."""
    from kgcn_amd import layers
    rng = np.random.default_rng(D)
    B, N = 9, 12
    x = (rng.standard_normal((B, N, D)) * 3 + 5 * rng.standard_normal(D)).astype(np.float32)    # per-feature offsets
    en = np.array([12, 3, 0, 7, 1, 12, 5, 9, 2]) if ragged else None
    gam = rng.standard_normal(D).astype(np.float32)
    bet = rng.standard_normal(D).astype(np.float32)
    gy = rng.standard_normal((B, N, D)).astype(np.float32)
    for phase in (0, 1):
        bn = layers.GraphBatchNormalization(learning_phase=phase)
        bn.build((B, N, D), dev())
        mm0 = rng.standard_normal(D).astype(np.float32)
        mv0 = rng.uniform(0.5, 2.0, D).astype(np.float32)
        with torch.no_grad():
            bn.gamma.copy_(t32(gam)); bn.beta.copy_(t32(bet))
            bn.moving_mean.copy_(t32(mm0)); bn.moving_variance.copy_(t32(mv0))
        tx = t32(x).requires_grad_(True)
        y = bn(tx, max_node_num=N, enabled_node_nums=en)
        ry, mean, var, nmm, nmv = K.graph_bn_fwd(x, gam, bet, mm0, mv0, en, training=bool(phase))
        close(y, ry, atol=2e-5, what="BN fwd phase %d" % phase)
        close(bn.moving_mean, nmm, atol=1e-5, what="moving mean phase %d" % phase)
        close(bn.moving_variance, nmv, atol=1e-5, rel=1e-6, what="moving variance phase %d" % phase)
        y.backward(t32(gy))
        dx, dg, db = K.graph_bn_bwd(x, gam, mean, var, gy, en, training=bool(phase))
        close(tx.grad, dx, atol=2e-5, what="BN dx phase %d" % phase)
        close(bn.gamma.grad, dg, atol=1e-5, rel=2e-6, what="BN dgamma phase %d" % phase)
        close(bn.beta.grad, db, atol=1e-5, rel=2e-6, what="BN dbeta phase %d" % phase)
        if en is not None:                                   # padding rows: exactly zero, forward and backward
            pad = np.arange(N)[None, :] >= en[:, None]
            assert float(y.detach()[torch.as_tensor(pad, device=dev())].abs().max()) == 0.0
            assert float(tx.grad[torch.as_tensor(pad, device=dev())].abs().max()) == 0.0
    # the module-level learning phase (K.set_learning_phase) is what a layer without its own setting follows, and the
    # reference's `training` keyword only freezes gamma / beta (it is Keras' `trainable`, layers.py:205)
    bn = layers.GraphBatchNormalization()
    try:
        layers.set_learning_phase(1)
        tx = t32(x).requires_grad_(True)
        y1 = bn(tx, enabled_node_nums=en, training=False)
        close(y1, K.graph_bn_fwd(x, np.ones(D), np.zeros(D), np.zeros(D), np.ones(D), en, training=True)[0], atol=2e-5)
        y1.sum().backward()
        assert bn.gamma.grad is None and tx.grad is not None
    finally:
        layers.set_learning_phase(0)
    close(bn(t32(x), enabled_node_nums=en),
          K.graph_bn_fwd(x, np.ones(D), np.zeros(D), bn.moving_mean.cpu().numpy(), bn.moving_variance.cpu().numpy(), en)[0],
          atol=2e-5, what="phase 0 uses the updated moving statistics")


@pytest.mark.parametrize("act", ["sigmoid", "relu", "tanh"])
@pytest.mark.parametrize("phase", [0, 1])
def test_graph_batch_normalization_fused_activation(act, phase):
    """act(bn(x)) in one pass (kgcn_graph_bn_apply_act_f32) and its backward with act'(y) applied while the gradient is
    read (kgcn_graph_bn_bwd_dact_f32) against the oracle's BN followed by the activation / its derivative; ragged batch,
    padding rows = act(0)."""
    from kgcn_amd import layers
    rng = np.random.default_rng(40 + phase)
    T, N, D = 37, 10, 50
    x = rng.standard_normal((T, N, D)).astype(np.float32) * 2 + 0.5
    en = rng.integers(0, N + 1, T).astype(np.int32)
    g = rng.standard_normal((T, N, D)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, D).astype(np.float32); beta = rng.standard_normal(D).astype(np.float32)
    mm = rng.standard_normal(D).astype(np.float32) * 0.3; mv = rng.uniform(0.5, 2.0, D).astype(np.float32)
    layer = layers.GraphBatchNormalization(learning_phase=phase, activation=act)
    layer.build(x.shape, dev())
    with torch.no_grad():
        layer.gamma.copy_(t32(gamma)); layer.beta.copy_(t32(beta))
        layer.moving_mean.copy_(t32(mm)); layer.moving_variance.copy_(t32(mv))
    tx = t32(x).requires_grad_(True)
    y = layer(tx, enabled_node_nums=torch.from_numpy(en).to(dev()))
    y.backward(t32(g))
    pre, mean, var, _, _ = K.graph_bn_fwd(x, gamma, beta, mm, mv, en, training=bool(phase))
    ref = _np_act(pre, act)
    close(y, ref, rel=2e-6, what="act(bn) " + act)
    aout = y.detach().cpu().numpy().astype(np.float64) if act == "relu" else ref
    dx, dgamma, dbeta = K.graph_bn_bwd(x, gamma, mean, var, g * _np_dact(aout, act), en, training=bool(phase))
    close(tx.grad, dx, rel=1e-5, what="dx")
    close(layer.gamma.grad, dgamma, rel=1e-5, atol=1e-4, what="dgamma")
    close(layer.beta.grad, dbeta, rel=1e-5, atol=1e-4, what="dbeta")


@pytest.mark.parametrize("frozen", ["gamma", "both"])
def test_inference_bn_backward_with_frozen_affine_inside_a_deferral_scope(frozen):
    """with learning phase 0 the BN backward queues the second stage of d gamma / d beta.  When gamma or
    beta does not require a gradient autograd drops that result as soon as backward() returns, and the flush would add D floats
    into whatever the caching allocator put there next.  The call must not defer then: a canary tensor allocated right after
    the backward (same size class as the dropped gradient) stays untouched by the flush, and dx / the wanted gradient are the
    ones of the plain call."""
    from kgcn_amd import layers, ops
    rng = np.random.default_rng(77)
    T, N, D = 64, 10, 50
    x = rng.standard_normal((T, N, D)).astype(np.float32)
    g = rng.standard_normal((T, N, D)).astype(np.float32)
    en = rng.integers(0, N + 1, T).astype(np.int32)
    layer = layers.GraphBatchNormalization(learning_phase=0)
    layer.build(x.shape, dev())
    with torch.no_grad():
        layer.gamma.copy_(t32(rng.uniform(0.5, 1.5, D).astype(np.float32)))
        layer.moving_variance.copy_(t32(rng.uniform(0.5, 2.0, D).astype(np.float32)))
    layer.gamma.requires_grad_(False)
    if frozen == "both":
        layer.beta.requires_grad_(False)
    ten = torch.from_numpy(en).to(dev())

    def run(defer):
        layer.beta.grad = None
        tx = t32(x).requires_grad_(True)
        y = layer(tx, enabled_node_nums=ten)
        cost = (y * t32(g)).sum()
        if defer:
            with ops.deferred_reductions(root=cost):
                cost.backward()
                canaries = [torch.full((D,), 7.0, device=dev()) for _ in range(8)]
                assert ops.lib.kgcn_reduce_pending() == 0, "the BN backward deferred a gradient nobody keeps"
            torch.cuda.synchronize()
            for c in canaries:
                assert bool((c == 7.0).all())
        else:
            cost.backward()
        torch.cuda.synchronize()
        return tx.grad.clone(), None if layer.beta.grad is None else layer.beta.grad.clone()

    dx0, db0 = run(False)
    dx1, db1 = run(True)
    assert torch.equal(dx0, dx1)
    assert (db0 is None) == (db1 is None) == (frozen == "both")
    if db0 is not None:
        assert torch.equal(db0, db1)
    gam = layer.gamma.detach().cpu().numpy()
    ref_dx = K.graph_bn_bwd(x, gam, layer.moving_mean.cpu().numpy(), layer.moving_variance.cpu().numpy(), g, en, training=False)[0]
    close(dx1, ref_dx, rel=1e-5, what="dx")


@pytest.mark.parametrize("channels,D", [("plain", 3), ("split", 50), ("norm", 64)])
def test_graph_maxpooling(channels, D):
    """kgcn/layers.py:122-153 (row N3): values on a coarse grid so that ties -- between entries and
    with the implicit zeros of the densified row -- really occur; fp32 oracle for exact tie sets."""
    from kgcn_amd import layers
    _, adjs = synthetic_batch("b30", channels)
    C = len(adjs[0])
    rng = np.random.default_rng(D + C)
    x = (rng.integers(-2, 3, size=(30, 10, D)) * 0.5).astype(np.float32)
    layer = layers.GraphMaxPooling(C)
    tx = t32(x).requires_grad_(True)
    out = layer(tx, adj=adjs)
    assert tuple(out.shape) == (30, 10, D) == tuple(layer.compute_output_shape((30, 10, D)))
    close(out, K.graph_maxpool_fwd(x, adjs, dtype=np.float32), atol=0, what="maxpool fwd")
    g = rng.standard_normal(x.shape).astype(np.float32)
    out.backward(t32(g))
    close(tx.grad, K.graph_maxpool_bwd(x, adjs, g, dtype=np.float32), atol=2e-6, what="maxpool bwd")
    x2 = rng.standard_normal((30, 10, D)).astype(np.float32)        # generic values, fp64 oracle
    t2 = t32(x2).requires_grad_(True)
    o2 = layer(t2, adj=adjs)
    close(o2, K.graph_maxpool_fwd(x2, adjs), what="maxpool fwd (random)")
    o2.backward(t32(g))
    close(t2.grad, K.graph_maxpool_bwd(x2, adjs, g), what="maxpool bwd (random)")


# ---------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ---------------------------------------------------------------------------------------------
def test_errors_are_loud():
    from kgcn_amd import BatchedCSR, ops, _lib
    with pytest.raises(ValueError):
        BatchedCSR.from_coo_list([(np.array([[0, 11]]), np.array([1.0]), [10, 10])], device=dev())
    csr = BatchedCSR.from_coo_list([(np.array([[0, 1]]), np.array([1.0]), [10, 10])], device=dev())
    with pytest.raises(_lib.KgcnHipError):
        ops.bspmm(csr, torch.zeros((10, 4)))               # CPU tensor: no CPU path
    with pytest.raises(_lib.KgcnHipError):
        ops.bspmm(csr, torch.zeros((12, 4), device=dev()))  # wrong number of rows
    with pytest.raises(_lib.KgcnHipError):
        ops.bspmm(csr, torch.zeros((10, 4), device=dev(), dtype=torch.float64))
    with pytest.raises(_lib.KgcnHipError):
        _lib.check(_lib.lib.kgcn_graphconv_fwd_f32(csr.padded4().desc(), None, None, None, 3, 80, None, None), "fused")
    assert b"not supported" in _lib.lib.kgcn_last_error()   # dout > 64
    with pytest.raises(_lib.KgcnHipError):                    # fused kernels want the row-padded layout
        _lib.check(_lib.lib.kgcn_graphconv_fwd_f32(csr.desc(), None, None, None, 4, 4, None, None), "fused")
    assert b"row_pad" in _lib.lib.kgcn_last_error()


# ---------------------------------------------------------------------------------------------
# BASELINE size (cfg2: 100k graphs x 32 nodes x 64 features): size-independent properties
# ---------------------------------------------------------------------------------------------
def test_cfg2_full_size_properties():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_cfg2
    from kgcn_amd import ops
    T = 100_000
    wl = make_cfg2(T, dev(), seed=1234)
    csr, x, w, b, g = wl["csr"], wl["x"], wl["w"], wl["bias"], wl["g"]
    assert csr.nnz == 100 * T and csr.max_nnz == 100
    # linearity of the aggregation: A(x + 2y) = Ax + 2Ay (bitwise-close in fp32)
    y = torch.roll(x, 1, 0)
    lhs = ops.bspmm(csr, x + 2 * y)
    rhs = ops.bspmm(csr, x) + 2 * ops.bspmm(csr, y)
    assert float((lhs - rhs).abs().max()) < 5e-5
    # adjacency is symmetric in this workload: A^T g == A g
    assert float((ops.bspmm(csr.transpose(), g) - ops.bspmm(csr, g)).abs().max()) < 1e-5
    # <A x, g> == <x, A^T g>  (adjoint identity, a checksum over the whole batch, fp64 accumulate)
    ax = ops.bspmm(csr, x)
    atg = ops.bspmm(csr.transpose(), g)
    l, r = float((ax.double() * g.double()).sum()), float((x.double() * atg.double()).sum())
    assert abs(l - r) <= 1e-6 * max(abs(l), abs(r), 1.0)
    # fused layer == dense + bspmm (two independent HIP implementations) at full size
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_f = ops.graphconv_fused(xg, wg, bg, csr)
    out_f.backward(g)
    xu, wu, bu = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_u = ops.bspmm(csr, ops.dense(xu.reshape(-1, 64), wu, bu)).reshape(T, 32, 64)
    out_u.backward(g)
    assert float((out_f - out_u).abs().max()) < 2e-5
    assert float((xg.grad - xu.grad).abs().max()) < 2e-5
    sw = float(wu.grad.abs().max())
    assert float((wg.grad - wu.grad).abs().max()) <= 2e-5 * sw
    assert float((bg.grad - bu.grad).abs().max()) <= 2e-5 * float(bu.grad.abs().max())
    # the WHOLE batch against the C restatement (oracle/kgcn_ref.c, fp32, OpenMP over graphs; about a second): out and dX
    # element-wise, dW / dbias -- sums over 3.2 M rows, where the cross-wave / cross-workgroup accumulation of the fused
    # backward could go wrong -- relative to their largest element (absolute 1e-5 is not meaningful for
    # sums over 1e5 graphs)
    from oracle import ref_c
    xh, gh, wh, bh = (t.detach().cpu().numpy() for t in (x, g, w, b))
    ro = ref_c.graphconv_fwd(wl["off"], wl["idx"], wl["val"], xh, wh, bh)
    rdx, rdw, rdb = ref_c.graphconv_bwd(wl["off"], wl["idx"], wl["val"], xh, wh, gh)
    close(out_f, ro, rel=2e-6, what="cfg2 full batch fwd vs C oracle")
    close(xg.grad, rdx, rel=2e-6, what="cfg2 full batch dX vs C oracle")
    # the C oracle itself accumulates dW in fp32 over 100k graphs (per-thread partials): compare both with an fp64
    # reduction of the oracle's per-graph products on a 4,096-graph prefix, and with each other on the whole batch
    close(wg.grad, rdw.reshape(64, 64), rel=2e-5, what="cfg2 full batch dW vs C oracle")
    close(bg.grad.reshape(-1), rdb.reshape(-1), rel=2e-5, what="cfg2 full batch dbias vs C oracle")
    Tp = 4096
    offp = wl["off"][:Tp + 1]
    nz = int(offp[-1])
    adjs_p = wl["adjs_of"](range(Tp))
    _, dw64, db64 = K.graphconv_bwd_fast(xh[:Tp], adjs_p, [wh], gh[:Tp])
    xq = x[:Tp].clone().requires_grad_(True)
    wq, bq = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    from kgcn_amd import BatchedCSR
    gi = np.repeat(np.arange(Tp), np.diff(offp))
    csr_p = BatchedCSR.from_arrays(gi, wl["idx"][:nz, 0].astype(np.int64), wl["idx"][:nz, 1].astype(np.int64),
                                   wl["val"][:nz], Tp, 32, 32, device=dev())
    ops.graphconv_fused(xq, wq, bq, csr_p).backward(g[:Tp])
    close(wq.grad, dw64[0], rel=2e-6, what="cfg2 4096-graph dW vs fp64 oracle")
    close(bq.grad.reshape(-1), db64[0].reshape(-1), rel=2e-6, what="cfg2 4096-graph dbias vs fp64 oracle")
    # a random sample of graphs against the fp64 oracle
    pick = np.random.default_rng(0).choice(T, 64, replace=False)
    adjs = wl["adjs_of"](pick)
    ref = K.graphconv_fwd_fast(x[pick].cpu().numpy(), adjs, [w.cpu().numpy()], [b.cpu().numpy()])
    close(out_f[pick], ref, rel=1e-6, what="cfg2 sample fwd")
    dxr, _, _ = K.graphconv_bwd_fast(x[pick].cpu().numpy(), adjs, [w.cpu().numpy()], g[pick].cpu().numpy())
    close(xg.grad[pick], dxr, rel=1e-6, what="cfg2 sample dX")


# ---------------------------------------------------------------------------------------------
# device-side mini-batch assembly
# ---------------------------------------------------------------------------------------------
def _same_container(a, b, what):
    assert (a.num_graphs, a.rows, a.cols, a.nnz, a.max_nnz, a.row_pad) == \
           (b.num_graphs, b.rows, b.cols, b.nnz, b.max_nnz, b.row_pad), what
    assert torch.equal(a.rowptr, b.rowptr), what + " rowptr"
    assert torch.equal(a.cv, b.cv), what + " cv"
    if a.row_pad:
        assert torch.equal(a.slots, b.slots), what + " slots"
        assert torch.equal(a.graph_ptr, b.graph_ptr), what + " graph_ptr"


@pytest.mark.parametrize("channels", ["plain", "split", "norm"])
def test_device_batch_assembly_bit_exact(channels):
    """kgcn_csr_gather_graphs vs the host assembly (FlatAdjacency.batch = kgcn/feed.py:112-126 restated):
    A, A^T and both row-padded containers of a shuffled, padded batch are identical arrays; a GraphConv
    step through either gives identical bits."""
    from kgcn_amd import data_util as D, layers
    z = load_golden("g1_synthetic_raw.npz")
    data = {"feature": z["feature"], "dense_adj": z["dense_adj"].astype(np.int64), "max_node_num": 10}
    chans, _ = D.build_adjs(data, normalize_adj_flag=(channels == "norm"), split_adj_flag=(channels == "split"))
    ds = D.DeviceGraphDataset(chans, z["feature"], device=dev())
    rng = np.random.default_rng(3)
    for idx, bs in [(rng.permutation(200)[:30], 30), (rng.permutation(200)[:10], 30), (np.arange(200), None),
                    (np.zeros(0, np.int64), 4)]:
        adj, feat = ds.batch(idx, bs)
        ref = D.batch_adjacency(chans, idx, bs, device=dev())
        reff = D.batch_features(z["feature"], idx, bs, device=dev())
        assert torch.equal(feat, reff)
        for c, (a, b) in enumerate(zip(adj.channels, ref.channels)):
            _same_container(a, b, "ch%d" % c)
            _same_container(a.transpose(), b.transpose(), "ch%d^T" % c)
            _same_container(a.padded4(), b.padded4(), "ch%d p4" % c)
            _same_container(a.transpose().padded4(), b.transpose().padded4(), "ch%d^T p4" % c)
        if len(idx):
            layer = layers.GraphConv(16, len(chans)).to(dev())
            x1 = feat.clone().requires_grad_(True)
            x2 = feat.clone().requires_grad_(True)
            o1, o2 = layer(x1, adj=adj), layer(x2, adj=ref)
            assert torch.equal(o1, o2)
            o1.sum().backward(); g1 = [p.grad.clone() for p in layer.parameters()]
            layer.zero_grad(); o2.sum().backward()
            assert torch.equal(x1.grad, x2.grad) and all(torch.equal(a, p.grad) for a, p in zip(g1, layer.parameters()))


@pytest.mark.parametrize("T,N,nnz,dups", [(40, 10, 300, True), (1, 32, 90, False), (500, 32, 40000, True), (7, 50, 0, False),
                                          (3, 300, 2000, True)])
def test_device_coo_pack_bit_exact(T, N, nnz, dups):
    """kgcn_coo_pack_f32 / kgcn_csr_pad4 vs the host packer (numpy stable sorts) on shuffled COO triples with duplicate
    entries: A, A^T and (N <= 32) both row-padded containers are identical arrays, the entry permutation too, and the
    kernels give identical bits through either."""
    from kgcn_amd import BatchedCSR, layers, BatchedAdjacency
    rng = np.random.default_rng(T * 1000 + nnz)
    g = rng.integers(0, T, nnz); r = rng.integers(0, N, nnz); c = rng.integers(0, N, nnz)
    if dups and nnz:
        k = nnz // 5
        g[:k], r[:k], c[:k] = g[-k:], r[-k:], c[-k:]                     # repeated (graph, row, col) triples
    v = rng.standard_normal(nnz).astype(np.float32)
    host = BatchedCSR.from_arrays(g, r, c, v, T, N, N, device=dev())
    ti = lambda a: torch.from_numpy(np.asarray(a, np.int32)).to(dev())
    devb = BatchedCSR.from_device_coo(ti(g), ti(r), ti(c), t32(v), T, N, N)
    _same_container(devb, host, "A")
    if nnz:
        order = np.argsort(g * N + r, kind="stable")
        assert np.array_equal(devb.perm.cpu().numpy(), order)
    _same_container(devb.transpose(), host.transpose(), "A^T")
    assert devb.transpose().transpose() is devb
    if N <= 32:
        _same_container(devb.padded4(), host.padded4(), "A p4")
        _same_container(devb.transpose().padded4(), host.transpose().padded4(), "A^T p4")
    if nnz:
        x = torch.randn(T, N, 16, device=dev())
        layer = layers.GraphConv(16, 1).to(dev())
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        o1, o2 = layer(x1, adj=BatchedAdjacency([devb])), layer(x2, adj=BatchedAdjacency([host]))
        assert torch.equal(o1, o2)
        o1.sum().backward(); g1 = [p.grad.clone() for p in layer.parameters()]
        layer.zero_grad(); o2.sum().backward()
        assert torch.equal(x1.grad, x2.grad) and all(torch.equal(a, p.grad) for a, p in zip(g1, layer.parameters()))
    ones = BatchedCSR.from_device_coo(ti(g), ti(r), ti(c), None, T, N, N)                 # val = NULL: all ones
    assert torch.equal(ones.cv[:, 0], host.cv[:, 0]) and bool((ones.cv[:, 1].view(torch.float32) == 1).all())


def test_coo_lists_take_the_device_packer_for_big_batches(monkeypatch):
    """The reference-API route (lists of per-graph COO, kgcn/feed.py:112-126) packs big batches on the GPU: same containers
    as the numpy route, bit for bit, and the same layer output."""
    from kgcn_amd import BatchedCSR, batched_csr, ops
    rng = np.random.default_rng(77)
    adjs = K.synth_mol_graphs(rng, 700, 32, 3)                       # ~70k entries > DEVICE_PACK_MIN_NNZ
    mats = [a[0] for a in adjs]
    devb = BatchedCSR.from_coo_list(mats, rows=32, cols=32, device=dev())
    assert devb._host is None, "expected the device packer"
    monkeypatch.setattr(batched_csr, "DEVICE_PACK_MIN_NNZ", 10 ** 12)
    host = BatchedCSR.from_coo_list(mats, rows=32, cols=32, device=dev())
    assert host._host is not None
    _same_container(devb, host, "A")
    _same_container(devb.transpose(), host.transpose(), "A^T")
    _same_container(devb.padded4(), host.padded4(), "A p4")
    _same_container(devb.transpose().padded4(), host.transpose().padded4(), "A^T p4")
    x = torch.randn(700, 32, 64, device=dev())
    w = torch.randn(64, 64, device=dev()); b = torch.randn(64, device=dev())
    assert torch.equal(ops.graphconv_fused(x, w, b, devb), ops.graphconv_fused(x, w, b, host))


def test_device_coo_pack_rejects_out_of_range_triples():
    from kgcn_amd import BatchedCSR
    ti = lambda a: torch.tensor(a, dtype=torch.int32, device=dev())
    with pytest.raises(ValueError, match="outside"):
        BatchedCSR.from_device_coo(ti([0, 1]), ti([0, 5]), ti([0, 0]), None, 2, 4, 4)
    with pytest.raises(ValueError, match="outside"):
        BatchedCSR.from_device_coo(ti([0, 2]), ti([0, 1]), ti([0, 0]), None, 2, 4, 4)


def test_device_batch_assembly_large_multiblock_scan():
    """70,000 selected graphs (> 256 x 256: the block-total scan loops) drawn with repetition from 3,000
    32-node graphs, every 7th a dummy; fused-kernel containers included."""
    from kgcn_amd import BatchedCSR
    rng = np.random.default_rng(12)
    G, N, T = 3000, 32, 70000
    adjs = K.synth_mol_graphs(rng, G, N, 3)
    src = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev())
    sel = rng.integers(0, G, size=T)
    sel[::7] = -1
    out = src.gather(sel)
    rp_src = src.rowptr.cpu().numpy().astype(np.int64)
    cv_src = src.cv.cpu().numpy()
    cnt = np.where(sel >= 0, rp_src[(np.maximum(sel, 0) + 1) * N] - rp_src[np.maximum(sel, 0) * N], 0)
    base = np.concatenate([[0], np.cumsum(cnt)])
    rows = (rp_src[:-1].reshape(G, N) - rp_src[:-1:N, None])[np.maximum(sel, 0)] * (sel >= 0)[:, None] + base[:-1, None]
    np.testing.assert_array_equal(out.rowptr.cpu().numpy(), np.concatenate([rows.reshape(-1), [base[-1]]]))
    pos = np.repeat(rp_src[np.maximum(sel, 0) * N] - base[:-1], cnt) + np.arange(base[-1])
    np.testing.assert_array_equal(out.cv.cpu().numpy(), cv_src[pos])
    p4 = out.padded4()
    gp = p4.graph_ptr.cpu().numpy()
    assert gp[0] == 0 and gp[-1] == p4.nnz and np.array_equal(gp, p4.rowptr.cpu().numpy()[::N])
    x = torch.randn(T, N, 64, device=dev())
    w = torch.randn(64, 64, device=dev()) * 0.1
    from kgcn_amd import ops
    o = ops.graphconv_fused(x, w, torch.zeros(1, 64, device=dev()), out)
    ref = ops.bspmm(out, ops.dense(x.reshape(T * N, 64), w, None).reshape(T, N, 64))
    close(o, ref.cpu().numpy(), atol=1e-4, what="fused on gathered batch")
    assert float(o[::7].abs().max()) == 0.0                                     # dummy graphs give zeros


# ---------------------------------------------------------------------------------------------
# d values through the layer + the integrated-gradients loop
# ---------------------------------------------------------------------------------------------
def test_integrated_gradients_features_and_adjacency():
    from kgcn_amd import BatchedAdjacency, layers, visualization
    rng = np.random.default_rng(19)
    B, N, F, Dh, D = 3, 10, 3, 8, 50
    adjs = K.normalize_adj(K.synth_mol_graphs(rng, B, N, 2))
    x = rng.standard_normal((B, N, F)).astype(np.float32)
    conv = layers.GraphConv(Dh, 1).to(dev())
    adj = BatchedAdjacency.from_adjs(adjs, n_nodes=N, device=dev())
    conv(t32(x), adj=adj)
    w, b = [conv.w[0].detach().cpu().numpy()], [conv.bias[0].detach().cpu().numpy()]
    ro = rng.standard_normal(Dh).astype(np.float32)
    tro = t32(ro)

    def score_fn(feat, a):
        return (layers.GraphGather()(torch.sigmoid(conv(feat, adj=a))) @ tro).sum()

    res = visualization.integrated_gradients(score_fn, t32(x), adj, divide_number=D)
    ig_x, ig_a = K.integrated_gradients(x, adjs, w, b, ro, D)
    close(res["features"], ig_x, atol=2e-6, what="IG features")
    # the reference's COO order is row-major here, = the CSR order of the container
    close(res["adjs"], np.concatenate(ig_a), atol=2e-6, what="IG adjacency values")
    assert abs(res["sum_of_ig"] - (res["end_score"] - res["start_score"])) < 0.05 * max(1.0, abs(res["end_score"]))
    dense = visualization.values_to_dense(adj.channels[0], res["adjs"])
    assert tuple(dense.shape) == (B, N, N) and abs(float(dense.sum()) - float(res["adjs"].sum())) < 1e-5
    one = visualization.integrated_gradients(score_fn, t32(x), adj, method="grad", modal=("adjs",))
    dx, dvals = K.probe_score_grads(x, adjs, w, b, ro)
    close(one["adjs"], np.concatenate([d[0] for d in dvals]), atol=2e-6, what="d score / d values")


@pytest.mark.parametrize("route", ["gathered", "static"])
def test_values_gradient_on_device_assembled_batches(route):
    """d values / d features through GraphConv when the batch was assembled on the device (DeviceGraphDataset.batch,
    StaticBatch): the backward runs A^T with the differentiable values permuted into A^T's entry order, which for a
    gathered container has to be derived on the device (BatchedCSR.transpose_perm).  ASYMMETRIC adjacency with unsorted
    columns, so a wrong permutation changes d features."""
    from kgcn_amd import layers, visualization
    from kgcn_amd.data_util import DeviceGraphDataset, FlatAdjacency
    rng = np.random.default_rng(23)
    G, N, F, Dh = 12, 10, 3, 8
    mats = []
    for _ in range(G):
        n = int(rng.integers(8, 25))
        idx = np.stack([rng.integers(0, N, n), rng.integers(0, N, n)], 1).astype(np.int32)   # duplicates allowed
        mats.append((idx, rng.standard_normal(n).astype(np.float32), [N, N]))
    feats = rng.standard_normal((G, N, F)).astype(np.float32)
    ds = DeviceGraphDataset([FlatAdjacency.from_coo_list(mats, n_nodes=N)], feats, device=dev())
    pick = np.array([7, 2, 2, 11, 0])
    if route == "gathered":
        adj, tx = ds.batch(pick, batch_size=6)                   # one dummy graph
    else:
        sb = ds.static_batch(6).load(pick)
        adj, tx = sb.adjacency, sb.features
    badjs = [[mats[i]] for i in pick] + [[(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [N, N])]]
    xb = np.concatenate([feats[pick], np.zeros((1, N, F), np.float32)])
    conv = layers.GraphConv(Dh, 1).to(dev())
    conv(tx, adj=adj)
    w, b = [conv.w[0].detach().cpu().numpy()], [conv.bias[0].detach().cpu().numpy()]
    ro = rng.standard_normal(Dh).astype(np.float32)
    tro = t32(ro)

    def score_fn(feat, a):
        return (layers.GraphGather()(torch.sigmoid(conv(feat, adj=a))) @ tro).sum()

    one = visualization.integrated_gradients(score_fn, tx, adj, method="grad")
    dx, dvals = K.probe_score_grads(xb, badjs, w, b, ro)
    close(one["features"], dx, atol=2e-6, what="d score / d features (%s batch)" % route)
    # the container stores a graph's entries in CSR order (stable by row); the oracle returns them in COO order
    got = one["adjs"].cpu().numpy()
    pos = 0
    for (idx, _, _), dv in zip([m[0] for m in badjs], dvals):
        order = np.argsort(np.asarray(idx).reshape(-1, 2)[:, 0], kind="stable")
        n = order.shape[0]
        np.testing.assert_allclose(got[pos:pos + n], np.asarray(dv[0])[order], rtol=0, atol=2e-6)
        pos += n
    # a static container is sized for the worst case: positions beyond the batch's entries belong to no graph
    assert pos == got.shape[0] if route == "gathered" else pos <= got.shape[0]


# ---------------------------------------------------------------------------------------------
# GAT (kgcn/layers.py:477-542)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("channels,D", [("plain", 3), ("split", 16), ("norm", 50), ("plain", 130)])
def test_gat_layer(channels, D):
    from kgcn_amd import layers
    _, adjs = synthetic_batch("b30", channels)              # 10 real + 20 dummy graphs: rows of sigmoid(0)
    C = len(adjs[0])
    rng = np.random.default_rng(D + C)
    x = (rng.standard_normal((30, 10, D)) * 0.7).astype(np.float32)
    layer = layers.GAT(C).to(dev())
    tx = t32(x).requires_grad_(True)
    out = layer(tx, adj=adjs)
    assert len(layer.weight_a) == C and tuple(layer.weight_a[0].shape) == (2 * D, 1)
    wa = [w.detach().cpu().numpy() for w in layer.weight_a]
    ref = K.gat_fwd(x, adjs, wa)
    close(out, ref, atol=2e-5, what="gat fwd")
    assert float(out[-1].min()) == 0.5 * C == float(out[-1].max())         # dummy graph
    g = rng.standard_normal(x.shape).astype(np.float32)
    out.backward(t32(g))
    dx, dwa = K.gat_bwd(x, adjs, wa, g)
    close(tx.grad, dx, atol=2e-5, rel=1e-5, what="gat dx")
    for c in range(C):
        close(layer.weight_a[c].grad, dwa[c], atol=2e-5, rel=2e-5, what="gat dweight_a[%d]" % c)


# ---------------------------------------------------------------------------------------------
# decoders and BatchGraphConv (kgcn/layers.py:268-397)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,D", [(30, 10, 50), (7, 50, 256), (300, 32, 64), (1, 3, 5)])
def test_decoders(B, N, D):
    from kgcn_amd import layers
    rng = np.random.default_rng(B + N + D)
    x = (rng.standard_normal((B, N, D)) * 0.5).astype(np.float32)
    g = rng.standard_normal((B, N, N)).astype(np.float32)
    tx = t32(x).requires_grad_(True)
    inner = layers.GraphDecoderInnerProd()
    out = inner(tx)
    assert tuple(out.shape) == (B, N, N) == tuple(inner.compute_output_shape((B, N, D)))
    close(out, K.gram_fwd(x), rel=2e-6, what="inner-product decoder")
    out.backward(t32(g))
    close(tx.grad, K.gram_bwd(x, None, g)[0], rel=5e-6, what="inner-product decoder dx")
    dm = layers.GraphDecoderDistMult().to(dev())
    tx2 = t32(x).requires_grad_(True)
    out2 = dm(tx2)
    w = dm.w[0].detach().cpu().numpy()
    assert w.shape == (D,)
    close(out2, K.gram_fwd(x, w), rel=2e-6, what="distmult decoder")
    out2.backward(t32(g))
    dx, dw = K.gram_bwd(x, w, g)
    close(tx2.grad, dx, rel=5e-6, what="distmult decoder dx")
    close(dm.w[0].grad, dw, rel=1e-5, what="distmult decoder dw")
    C = 3
    dist = layers.DistMult(adj_channel_num=C).to(dev())
    tx3 = t32(x).requires_grad_(True)
    out3 = dist(tx3)
    assert tuple(out3.shape) == (B, C, N, N) and tuple(dist.w[0].shape) == (C, D)
    ww = dist.w[0].detach().cpu().numpy()
    for c in range(C):
        close(out3[:, c], K.gram_fwd(x, ww[c]), rel=2e-6, what="DistMult channel %d" % c)
    g4 = rng.standard_normal((B, C, N, N)).astype(np.float32)
    out3.backward(t32(g4))
    dxs = sum(K.gram_bwd(x, ww[c], g4[:, c])[0] for c in range(C))
    close(tx3.grad, dxs, rel=5e-6, what="DistMult dx")
    close(dist.w[0].grad, np.stack([K.gram_bwd(x, ww[c], g4[:, c])[1] for c in range(C)]), rel=1e-5, what="DistMult dw")
    l1, l2 = t32(x[:, 0]), t32(x[:, 1])
    close(dist.compute_score(l1, l2, 1), (x[:, 0] * x[:, 1] * ww[1]).sum(1), rel=2e-6, what="DistMult score")
    close(dist.compute_left_prediction(t32(x[0]), l2, 2), (x[:, 1] * ww[2]) @ x[0].T, rel=2e-6, what="left prediction")
    close(dist.compute_right_prediction(l1, t32(x), 0), np.einsum("bnd,bd->bn", x, x[:, 0] * ww[0]), rel=2e-6,
          what="right prediction")


def test_batch_graphconv_block_diagonal():
    from kgcn_amd import layers
    rng = np.random.default_rng(12)
    G, N, F, Dout = 16, 20, 24, 40
    adjs = K.synth_mol_graphs(rng, G, N, 2, normalize=True)
    big = K.block_diag_csr(adjs, 0, N).tocoo()
    coo = (np.stack([big.row, big.col], 1).astype(np.int64), big.data.astype(np.float32), [G * N, G * N])
    net = rng.standard_normal((G * N, F)).astype(np.float32)
    torch.manual_seed(12)                   # the layer's initialiser draws from torch's generator: independent of the tests run before
    layer = layers.BatchGraphConv(Dout)
    tn = t32(net).requires_grad_(True)
    out = layer([tn, coo])
    with torch.no_grad():
        layer.bias.copy_(t32(rng.standard_normal(Dout) * 0.1))
    out = layer([tn, coo])
    assert tuple(out.shape) == (G * N, Dout) and tuple(layer.bias.shape) == (Dout,)
    ref = K.batch_graphconv_fwd(net, coo, layer.w.detach().cpu().numpy(), layer.bias.detach().cpu().numpy())
    close(out, ref, rel=2e-6, what="BatchGraphConv")
    # backward: relu mask on the aggregated pre-activation, then the GraphConv gradients (fp64 restatement)
    gy = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(t32(gy))
    w64, b64 = layer.w.detach().cpu().numpy().astype(np.float64), layer.bias.detach().cpu().numpy().astype(np.float64)
    gm = gy.astype(np.float64) * (ref > 0)
    dfw = K.spmm_coo(coo, gm, adjoint_a=True)
    close(tn.grad, dfw @ w64.T, rel=2e-6, what="BatchGraphConv d net")
    close(layer.w.grad, net.astype(np.float64).T @ dfw, rel=2e-6, what="BatchGraphConv d kernel")
    close(layer.bias.grad, dfw.sum(0), rel=2e-6, what="BatchGraphConv d bias")
    # no pre-activation may sit so close to the relu kink that fp32 and fp64 disagree on the mask
    # (a guard on the drawn data, checked AFTER the comparisons: an element nearer than an fp32 rounding of the pre-activation could flip)
    pre = K.spmm_coo(coo, net.astype(np.float64) @ w64 + b64.reshape(1, -1))
    assert np.abs(pre).min() > 2e-7
