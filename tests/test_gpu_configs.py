"""Parity at the shapes of BASELINE configs 3-5 (the generic kernels: fp32-MFMA dense contraction
+ batched SpMM), forward and backward against the oracle, through the layer API.

cfg3  example_config/sparse.json path: ONE block-diagonal [sumN x sumN] adjacency, batch size 1
      (kgcn/data_util.py:698-845, example_model/sparse.py:65-69), 128 molecules x 50 nodes,
      128-dim features, Kipf-normalised values.
cfg4  Tox21-shaped multitask model (example_model/model_multitask.py:51-57): N = 50 padded with
      variable true size, F = 81, GraphConv 256 -> GraphDense 256 -> GraphConv 50, 12 tasks.
cfg5  GIN (example_model/model_gin.py:44-54) on ring graphs (data_generator/
      synth_generator_ring.py), N = 10, 256-dim features.
"""
import numpy as np
import pytest
import torch

from oracle import kgcn_oracle as K
from test_gpu_parity import close, dev, t32

pytestmark = pytest.mark.gpu


def test_cfg3_block_diagonal_graphconv():
    from kgcn_amd import BatchedAdjacency, BatchedCSR, layers
    rng = np.random.default_rng(33)
    G, N, F, Dout = 128, 50, 128, 128
    adjs = K.synth_mol_graphs(rng, G, N, 8, normalize=True)
    big = K.block_diag_csr(adjs, 0, N).tocoo()
    csr = BatchedCSR.from_arrays(np.zeros(big.nnz, np.int64), big.row, big.col, big.data, 1, G * N, G * N,
                                 device=dev())
    x = rng.standard_normal((1, G * N, F)).astype(np.float32)       # net[None] of sparse.py:69
    layer = layers.GraphConv(Dout, 1)
    tx = t32(x).requires_grad_(True)
    out = layer(tx, BatchedAdjacency([csr]))                          # positional adj, batch of ONE
    assert tuple(out.shape) == (1, G * N, Dout)
    w = [layer.w[0].detach().cpu().numpy()]
    with torch.no_grad():
        layer.bias[0].copy_(t32(rng.standard_normal((1, Dout))))
    b = [layer.bias[0].detach().cpu().numpy()]
    out = layer(tx, BatchedAdjacency([csr]))
    xg = x.reshape(G, N, F)
    ref = K.graphconv_fwd_fast(xg, adjs, w, b).reshape(1, G * N, Dout)
    close(out, ref, rel=2e-6, what="cfg3 fwd")
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(t32(g))
    dx, dw, db = K.graphconv_bwd_fast(xg, adjs, w, g.reshape(G, N, Dout))
    close(tx.grad, dx.reshape(1, G * N, F), rel=2e-6, what="cfg3 dX")
    close(layer.w[0].grad, dw[0], rel=1e-5, what="cfg3 dW")
    close(layer.bias[0].grad, db[0], rel=1e-5, what="cfg3 db")


def test_cfg4_multitask_shaped_stack():
    from kgcn_amd import layers
    rng = np.random.default_rng(44)
    B, N, F, T = 96, 50, 81, 12
    sizes = rng.integers(5, 51, size=B)
    adjs = []
    for n in sizes:                                       # molecule of n nodes padded to N = 50
        a = K.synth_mol_graphs(rng, 1, int(n), 2)[0][0]
        adjs.append([(a[0], a[1], [N, N])])
    adjs = K.normalize_adj(adjs)
    x = np.zeros((B, N, F), np.float32)
    for b_, n in enumerate(sizes):
        x[b_, :n] = rng.standard_normal((n, F))
    l1, l2, l3 = layers.GraphConv(256, 1), layers.GraphDense(256), layers.GraphConv(50, 1)
    tx = t32(x).requires_grad_(True)
    h1 = l1(tx, adj=adjs)
    h2 = l2(torch.relu(h1))
    h3 = l3(torch.relu(h2), adj=adjs)
    pooled = layers.GraphGather()(torch.relu(h3))
    assert tuple(pooled.shape) == (B, 50)
    relu = lambda a: np.maximum(a, 0)
    p = lambda t: t.detach().cpu().numpy()
    r1 = K.graphconv_fwd_fast(x, adjs, [p(l1.w[0])], [p(l1.bias[0])])
    close(h1, r1, rel=2e-6, what="cfg4 conv1")
    r2 = K.graphdense_fwd(relu(r1), p(l2.kernel), p(l2.bias))
    close(h2, r2, rel=2e-6, what="cfg4 dense")
    r3 = K.graphconv_fwd_fast(relu(r2), adjs, [p(l3.w[0])], [p(l3.bias[0])])
    close(h3, r3, rel=2e-6, what="cfg4 conv2")
    close(pooled, K.gather_fwd(relu(r3)), rel=2e-6, what="cfg4 gather")
    # backward through the whole stack against the oracle's chain rule
    g = rng.standard_normal((B, 50)).astype(np.float32)
    pooled.backward(t32(g))
    # relu masks are taken from the GPU activations: a pre-activation of ~1e-9 may have a different
    # sign in fp32 and in the fp64 oracle, which would flip one gradient mask (not a kernel error)
    m1, m2, m3 = p(h1) > 0, p(h2) > 0, p(h3) > 0
    d3 = K.gather_bwd(g, N) * m3
    dx3, dw3, db3 = K.graphconv_bwd_fast(relu(r2), adjs, [p(l3.w[0])], d3)
    close(l3.w[0].grad, dw3[0], rel=1e-5, what="cfg4 dW3")
    d2 = dx3 * m2
    dx2, dk2, dbias2 = K.graphdense_bwd(relu(r1), p(l2.kernel), d2)
    close(l2.kernel.grad, dk2, rel=1e-5, what="cfg4 dK")
    close(l2.bias.grad, dbias2, rel=1e-5, what="cfg4 dbias")
    d1 = dx2 * m1
    dx1, dw1, db1 = K.graphconv_bwd_fast(x, adjs, [p(l1.w[0])], d1)
    close(l1.w[0].grad, dw1[0], rel=1e-5, what="cfg4 dW1")
    close(l1.bias[0].grad, db1[0], rel=1e-5, what="cfg4 db1")
    close(tx.grad, dx1, rel=1e-5, what="cfg4 dX")


def test_cfg5_gin_ring_graphs_256():
    from kgcn_amd import layers
    rng = np.random.default_rng(55)
    B, N, D = 200, 10, 256
    adjs = K.synth_ring_graphs(rng, B, N)
    x = rng.standard_normal((B, N, D)).astype(np.float32)
    gin, d1, d2 = layers.GINAggregate(1), layers.GraphDense(256), layers.GraphDense(256)
    tx = t32(x).requires_grad_(True)
    a = gin(tx, adj=adjs)
    with torch.no_grad():
        gin.epsilon[0].fill_(0.25)
    a = gin(tx, adj=adjs)
    z1 = d1(a)
    z2 = d2(torch.relu(z1))
    h = torch.relu(z2)
    out = layers.GraphGather()(h)
    p = lambda t: t.detach().cpu().numpy()
    relu = lambda v: np.maximum(v, 0)
    ra = K.gin_fwd(x, adjs, [0.25])
    close(a, ra, rel=2e-6, what="cfg5 gin")
    r1 = K.graphdense_fwd(ra, p(d1.kernel), p(d1.bias))
    r2 = K.graphdense_fwd(relu(r1), p(d2.kernel), p(d2.bias))
    close(out, K.gather_fwd(relu(r2)), rel=2e-6, what="cfg5 readout")
    g = rng.standard_normal((B, 256)).astype(np.float32)
    out.backward(t32(g))
    mz1, mz2 = p(z1) > 0, p(z2) > 0                 # masks from the GPU activations (see cfg4)
    dg2 = K.gather_bwd(g, N) * mz2
    dxx2, dk2, _ = K.graphdense_bwd(relu(r1), p(d2.kernel), dg2)
    close(d2.kernel.grad, dk2, rel=1e-5, what="cfg5 dK2")
    dxx1, dk1, _ = K.graphdense_bwd(ra, p(d1.kernel), dxx2 * mz1)
    close(d1.kernel.grad, dk1, rel=1e-5, what="cfg5 dK1")
    dxa, deps = K.gin_bwd(x, adjs, [0.25], dxx1)
    close(tx.grad, dxa, rel=1e-5, what="cfg5 dX")
    # d eps = <dA, x>: a signed sum of B*N*D = 512k products whose terms cancel (weights are random per
    # run, |sum| can be far below the sum of |terms|): fp32 tolerance relative to sum |dA . x|, like any dot
    scale = float(np.abs(dxx1.astype(np.float64) * x).sum())
    assert abs(float(gin.epsilon[0].grad) - float(deps[0])) <= 2e-6 * scale, (float(gin.epsilon[0].grad), deps[0], scale)
