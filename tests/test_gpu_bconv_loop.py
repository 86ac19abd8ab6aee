"""The channel-loop Bconv kernels (round 6: bconv_loop_kernel behind kgcn_bconv_act_f32, bconv_fanout_kernel behind
kgcn_bconv_fanout_f32; kgcn/bconv_call.py:11-23 forward, :45-53 the gradient's fan-out) against the numpy oracle and, bit for
bit, against the single-channel launches they replace -- shapes that take the new route (N <= 64, widths of whole 16- / 8-byte
vectors), shapes that fall back (odd width, N > 64), every activation code, empty graphs / empty channels, more than 128 entries
in one channel of one graph."""
import numpy as np
import pytest
import torch

from test_gpu_parity import K, close, dev, t32

pytestmark = pytest.mark.gpu
ACT = {None: 0, "sigmoid": 1, "relu": 2, "tanh": 3}


def _channels(rng, T, N, C, density, empty_graph_every=0, dense_channel=None):
    """adjs[b][c] = (indices [nnz, 2], values, [N, N]); unsorted, with duplicates"""
    adjs = []
    for b in range(T):
        row = []
        for c in range(C):
            dens = 0.9 if dense_channel == c else density
            a = (rng.random((N, N)) < dens) * rng.standard_normal((N, N))
            if empty_graph_every and b % empty_graph_every == 1:
                a[:] = 0
            if c == C - 1 and b % 3 == 0:
                a[:] = 0                                           # an empty channel of this graph
            idx, val, shp = K.dense_to_sparse(a)
            idx = np.asarray(idx).reshape(-1, 2).astype(np.int32)
            val = np.asarray(val, np.float32)
            if len(val) > 2:
                idx = np.concatenate([idx, idx[:2]]); val = np.concatenate([val, val[:2]])      # duplicates accumulate
                p = rng.permutation(len(val)); idx, val = idx[p], val[p]
            row.append((idx, val, [N, N]))
        adjs.append(row)
    return adjs


def _act(v, act):
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))
    if act == "relu":
        return np.maximum(v, 0.0)
    if act == "tanh":
        return np.tanh(v)
    return v


def _dact(a, act):
    if act == "sigmoid":
        return a * (1.0 - a)
    if act == "relu":
        return (a > 0).astype(np.float64)
    if act == "tanh":
        return 1.0 - a * a
    return np.ones_like(a)


@pytest.mark.parametrize("T,N,C,d,act,dense_channel", [
    (37, 10, 6, 50, "sigmoid", None),       # synthetic.jbl's split_adj shape: 8-byte vectors, 25 lanes per row
    (65, 32, 6, 64, None, None),            # cfg2's shape with six channels
    (20, 32, 3, 64, "relu", 1),             # one channel with > 128 entries per graph (the slice beyond the registers)
    (9, 50, 2, 256, "tanh", None),          # wide operand: column slices of 64
    (12, 64, 8, 32, "sigmoid", None),       # the largest graph and channel count one launch takes
    (7, 10, 6, 51, "relu", None),           # odd width: the old kernels
    (3, 70, 2, 64, None, None),             # N > 64: the old kernels
    (5, 17, 9, 24, "sigmoid", None),        # more channels than one launch takes: groups + accumulation
])
def test_bconv_forward_and_fanout(T, N, C, d, act, dense_channel):
    from kgcn_amd import ops
    from kgcn_amd._lib import lib, ptr, current_stream, check
    from kgcn_amd.batched_csr import BatchedAdjacency
    rng = np.random.default_rng(T * 1000 + N * 10 + C)
    adjs = _channels(rng, T, N, C, 0.12, empty_graph_every=5, dense_channel=dense_channel)
    adj = BatchedAdjacency.from_adjs(adjs, n_nodes=N, device=dev())
    rhs = rng.standard_normal((T * N, C * d)).astype(np.float32)
    g = rng.standard_normal((T * N, d)).astype(np.float32)
    dense = [[rhs[b * N:(b + 1) * N, c * d:(c + 1) * d] for c in range(C)] for b in range(T)]
    ref = np.stack(K.bconv(adjs, dense)).astype(np.float64)
    ref_act = _act(ref, act)
    tr = t32(rhs).requires_grad_(True)
    out = ops.bconv(adj, tr, d, activation=act)
    close(out, ref_act.reshape(T * N, d), rel=2e-6, what="bconv forward (%s)" % act)
    out.backward(t32(g))
    # d rhs_c = A_c^T (g (.) act'(out))
    aout = out.detach().cpu().numpy().astype(np.float64) if act == "relu" else ref_act.reshape(T * N, d)
    dpre = (g.astype(np.float64) * _dact(aout, act)).reshape(T, N, d)
    _, rg = K.bconv_grad(adjs, dense, list(dpre))
    want = np.concatenate([np.concatenate([rg[b][c] for c in range(C)], axis=1) for b in range(T)], axis=0)
    close(tr.grad, want, rel=2e-6, what="bconv fan-out adjoint (%s)" % act)
    # bit for bit against the launches the fan-out replaces: one kgcn_bspmm_dact_f32 per channel
    tg, ta = t32(g), out.detach()
    one = torch.empty((T * N, C * d), device=dev())
    for c, ch in enumerate(adj.channels):
        check(lib.kgcn_bspmm_dact_f32(ch.transpose().desc(), ptr(tg), ptr(ta) if act else None, d, N * d, d, ACT[act],
                                      one.data_ptr() + 4 * c * d, C * d, N * C * d, 0.0, current_stream()), "kgcn_bspmm_dact_f32")
    assert torch.equal(one, tr.grad), "fan-out differs from the per-channel launches"


def test_fanout_argument_checks():
    from kgcn_amd._lib import lib, ptr, current_stream
    from kgcn_amd.batched_csr import BatchedAdjacency
    rng = np.random.default_rng(1)
    adjs = _channels(rng, 4, 10, 2, 0.2)
    adj = BatchedAdjacency.from_adjs(adjs, n_nodes=10, device=dev())
    g = torch.zeros((40, 8), device=dev()); o = torch.zeros((40, 16), device=dev())
    at = adj.desc_array(True)
    assert lib.kgcn_bconv_fanout_f32(at, 0, ptr(g), None, 8, 80, 8, 0, ptr(o), 16, 160, 8, current_stream()) != 0
    assert lib.kgcn_bconv_fanout_f32(at, 2, None, None, 8, 80, 8, 0, ptr(o), 16, 160, 8, current_stream()) != 0
    assert lib.kgcn_bconv_fanout_f32(at, 2, ptr(g), None, 8, 80, 8, 1, ptr(o), 16, 160, 8, current_stream()) != 0     # act without act_out
    assert lib.kgcn_bconv_fanout_f32(at, 2, ptr(g), None, 4, 80, 8, 0, ptr(o), 16, 160, 8, current_stream()) != 0     # ld < d
    assert lib.kgcn_bconv_fanout_f32(at, 2, ptr(g), None, 8, 80, 8, 0, ptr(o), 16, 160, 8, current_stream()) == 0


@pytest.mark.parametrize("T,N,C,din,dout,route", [
    (200, 10, 3, 5, 32, "aggregate-first"),      # din + 1 < dout, >= 1,024 rows: [A_0 X' | A_1 X' | ...] [W_0; b_0; 0; ...]
    (120, 10, 6, 3, 50, "aggregate-first"),      # model.py's first layer with split adjacency
    (64, 32, 6, 64, 64, "contract-first"),       # one GEMM over [W_0 | W_1 | ...] + the channel-loop Bconv
    (30, 10, 6, 50, 50, "contract-first"),       # synthetic.jbl's batch with split adjacency (below 1,024 rows)
])
def test_multichannel_graphconv_layer_without_torch_glue(T, N, C, din, dout, route):
    """The multi-channel GraphConv layer on both of its routes against the oracle (kgcn/layers.py:64-116) -- and no torch operator
    may launch anything inside its forward + backward (the two torch.cat of the C kernels / biases and the 2 C gradient clones
    they caused are one launch each way now: kgcn_copy2d_multi_f32)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import aten_in_step
    from kgcn_amd import layers
    from kgcn_amd.batched_csr import BatchedAdjacency
    rng = np.random.default_rng(T + N + C + din)
    adjs = _channels(rng, T, N, C, 0.15, empty_graph_every=7)
    adj = BatchedAdjacency.from_adjs(adjs, n_nodes=N, device=dev())
    x = rng.standard_normal((T, N, din)).astype(np.float32)
    g = rng.standard_normal((T, N, dout)).astype(np.float32)
    layer = layers.GraphConv(dout, C).to(dev())
    layer.build((T, N, din), dev())
    with torch.no_grad():
        for b in layer.bias:
            b.copy_(t32(rng.standard_normal((1, dout)).astype(np.float32) * 0.1))
    assert (route == "aggregate-first") == (layers.aggregate_first and (din + 1 + 3) // 4 * 4 < dout and T * N >= 1024)
    tx = t32(x).requires_grad_(True)
    tg = t32(g)
    out = layer(tx, adj=adj)
    out.backward(tg)
    w = [p.detach().cpu().numpy() for p in layer.w]
    b = [p.detach().cpu().numpy() for p in layer.bias]
    close(out, K.graphconv_fwd(x, adjs, w, b), rel=2e-6, what="multi-channel GraphConv forward (%s)" % route)
    dx, dw, db = K.graphconv_bwd(x, adjs, w, b, g)
    close(tx.grad, dx, rel=2e-6, what="d inputs")
    for c in range(C):
        close(layer.w[c].grad, dw[c], rel=1e-5, what="d kernel%d" % c)
        close(layer.bias[c].grad, db[c].reshape(1, dout), rel=1e-5, what="d bias%d" % c)

    def step():
        tx.grad = None
        for p in layer.parameters():
            p.grad = None
        layer(tx, adj=adj).backward(tg)

    step()
    seen = aten_in_step.log_step(step)
    assert not seen, "torch operators inside the multi-channel layer: %s" % list(seen.items())
