"""Every layer of the path at batch sizes beyond every launch cap of its kernels, through a size-independent property: graphs
(and node rows) are independent, so op(whole batch) restricted to a slice of graphs == op(that slice), and the parameter
gradients of the whole batch == the sum over the slices.  The slices are small enough for ONE trip of every persistent kernel
(<= 30,000 node rows: at most one 64-row tile per workgroup slot) -- the sizes at which the same ops are checked against the
oracle (test_gpu_parity.py, test_gpu_model.py); the whole batches are 600,000 node rows.  What this catches: per-tile state in
workgroups that walk several tiles, partial sums spread over more workgroups than a small case launches, 32-bit offsets.
Training-mode GraphBatchNormalization (statistics over the batch), the losses and TF-Adam are compared with closed forms."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_parity import close, dev, t32, _set_variant  # noqa: E402

pytestmark = pytest.mark.gpu

B_BIG, N_NODES, CHUNK = 60_000, 10, 3_000


def _channels(B, C, seed):
    """C adjacency channels of B ring graphs (bench.gen_ring_graphs) with random values, as flat COO arrays."""
    import bench
    out = []
    for c in range(C):
        g, r, col, _, rng = bench.gen_ring_graphs(B, N_NODES, seed=seed + c)
        out.append((g, r, col, (rng.random(g.shape[0]) + 0.5).astype(np.float32)))
    return out


def _adjacency(chans, lo, hi):
    from kgcn_amd import BatchedAdjacency, BatchedCSR
    cs = []
    for g, r, c, v in chans:
        a, b = np.searchsorted(g, lo), np.searchsorted(g, hi)
        cs.append(BatchedCSR.from_arrays(g[a:b] - lo, r[a:b], c[a:b], v[a:b], hi - lo, N_NODES, N_NODES, device=dev()))
    return BatchedAdjacency(cs)


def _run(layer, x, up, adj=None, **kw):
    for p in layer.parameters():
        p.grad = None
    tx = x.clone().requires_grad_(True)
    out = layer(tx, adj=adj, **kw) if adj is not None else layer(tx, **kw)
    out.backward(up)
    return out.detach(), tx.grad, [p.grad.clone() for p in layer.parameters()]


def _check_split(layer, x, up, chans=None, what="", rel_out=2e-6, rel_par=2e-5, **kw):
    B = x.shape[0]
    adj = _adjacency(chans, 0, B) if chans is not None else None
    out, dx, pg = _run(layer, x, up, adj, **kw)
    so, sx = float(out.abs().max()), float(dx.abs().max())
    acc = [torch.zeros_like(g, dtype=torch.float64) for g in pg]
    for lo in range(0, B, CHUNK):
        hi = min(B, lo + CHUNK)
        a = _adjacency(chans, lo, hi) if chans is not None else None
        kc = {k: (v[lo:hi] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in kw.items()}
        o, d, g = _run(layer, x[lo:hi], up[lo:hi], a, **kc)
        eo, ed = float((out[lo:hi] - o).abs().max()), float((dx[lo:hi] - d).abs().max())
        assert eo <= rel_out * so, "%s: outputs of graphs %d..%d differ by %.3e (scale %.3e)" % (what, lo, hi, eo, so)
        assert ed <= rel_out * 4 * sx, "%s: d inputs of graphs %d..%d differ by %.3e (scale %.3e)" % (what, lo, hi, ed, sx)
        for s, gi in zip(acc, g):
            s += gi.double()
    for (n_, _), whole, parts in zip(layer.named_parameters(), pg, acc):
        sc = float(parts.abs().max())
        e = float((whole.double() - parts).abs().max())
        assert e <= rel_par * sc + 1e-30, "%s: grad %s of the whole batch differs from the sum over slices by %.3e (scale %.3e)" % (what, n_, e, sc)


def _features(B, D, seed):
    g = torch.Generator(device=dev()); g.manual_seed(seed)
    return torch.randn((B, N_NODES, D), device=dev(), generator=g)


@pytest.mark.parametrize("variant", ["default", "bspmm", "bconv", "batched"])
@pytest.mark.parametrize("din,dout,act", [(16, 50, "sigmoid"), (64, 64, None)])
def test_graphconv_two_channels_whole_batch_equals_its_slices(variant, din, dout, act):
    from kgcn_amd import layers
    chans = _channels(B_BIG, 2, seed=11)
    x, up = _features(B_BIG, din, 1), _features(B_BIG, dout, 2)
    try:
        _set_variant(variant)
        torch.manual_seed(3)
        layer = layers.GraphConv(dout, 2, activation=act)
        layer.build((B_BIG, N_NODES, din), dev())
        with torch.no_grad():
            for b in layer.bias:
                b.normal_(0, 0.2)
        _check_split(layer, x, up, chans, what="GraphConv[%s] %d->%d" % (variant, din, dout))
    finally:
        _set_variant("default")


def test_graphconv_one_channel_fused_kernels_whole_batch_equals_its_slices():
    """C = 1, N <= 32, D <= 64: the fused forward / backward kernels of bench.py's headline path, 10-node graphs."""
    from kgcn_amd import layers
    chans = _channels(B_BIG, 1, seed=5)
    x, up = _features(B_BIG, 64, 4), _features(B_BIG, 64, 5)
    torch.manual_seed(3)
    layer = layers.GraphConv(64, 1, activation="tanh")
    layer.build((B_BIG, N_NODES, 64), dev())
    _check_split(layer, x, up, chans, what="GraphConv fused 64->64")


def test_gin_gat_maxpool_gather_whole_batch_equals_their_slices():
    from kgcn_amd import layers
    chans = _channels(B_BIG, 2, seed=21)
    x, up = _features(B_BIG, 48, 6), _features(B_BIG, 48, 7)
    gin = layers.GINAggregate(2)
    gin.build((B_BIG, N_NODES, 48), dev())
    with torch.no_grad():
        gin.epsilon[0].fill_(0.25); gin.epsilon[1].fill_(-0.5)
    _check_split(gin, x, up, chans, what="GINAggregate", rel_par=5e-5)
    torch.manual_seed(1)
    gat = layers.GAT(2)
    gat.build((B_BIG, N_NODES, 48), dev())
    _check_split(gat, x, up, chans, what="GAT", rel_out=5e-6, rel_par=5e-5)
    _check_split(layers.GraphMaxPooling(2), x, up, chans, what="GraphMaxPooling")
    g = torch.Generator(device=dev()); g.manual_seed(9)
    _check_split(layers.GraphGather(), x, torch.randn((B_BIG, 48), device=dev(), generator=g), what="GraphGather")


def test_decoders_whole_batch_equals_their_slices():
    """GraphDecoderInnerProd / GraphDecoderDistMult (kgcn/layers.py:268-305): adj_hat[b] = (w *) X[b] X[b]^T of 60,000 graphs."""
    from kgcn_amd import layers
    x = _features(B_BIG, 24, 14)
    g = torch.Generator(device=dev()); g.manual_seed(15)
    up = torch.randn((B_BIG, N_NODES, N_NODES), device=dev(), generator=g)
    _check_split(layers.GraphDecoderInnerProd(), x, up, what="GraphDecoderInnerProd")
    torch.manual_seed(4)
    dm = layers.GraphDecoderDistMult()
    dm.build((B_BIG, N_NODES, 24), dev())
    _check_split(dm, x, up, what="GraphDecoderDistMult", rel_par=5e-5)


@pytest.mark.parametrize("din,dout,act", [(3, 50, "sigmoid"), (50, 50, "relu"), (64, 64, None), (81, 256, "sigmoid"), (256, 50, None),
                                          (50, 256, "tanh"), (256, 12, None), (12, 256, "relu"), (256, 256, "relu"), (512, 256, None),
                                          (128, 128, "sigmoid"), (256, 2, None)])
def test_graphdense_whole_batch_equals_its_slices(din, dout, act):
    from kgcn_amd import layers
    B = B_BIG if din * dout <= 65536 else B_BIG // 2
    x, up = _features(B, din, 8), _features(B, dout, 9)
    torch.manual_seed(2)
    layer = layers.GraphDense(dout, activation=act)
    layer.build((B, N_NODES, din), dev())
    with torch.no_grad():
        layer.bias.normal_(0, 0.2)
    _check_split(layer, x, up, what="GraphDense %d->%d %s" % (din, dout, act))


def test_dense_read_out_fused_with_gather_whole_batch_equals_its_slices():
    """ops.dense_gather (model_gin.py's block output: GraphDense + GraphGather, the gradient joined inside the dX GEMM)."""
    from kgcn_amd import layers
    x = _features(B_BIG // 2, 256, 10)
    B = x.shape[0]
    g = torch.Generator(device=dev()); g.manual_seed(3)
    up_y, up_p = torch.randn((B, N_NODES, 256), device=dev(), generator=g), torch.randn((B, 256), device=dev(), generator=g)
    torch.manual_seed(2)
    dense = layers.GraphDense(256, activation="relu")
    dense.build((B, N_NODES, 256), dev())

    def run(lo, hi):
        dense.kernel.grad = dense.bias.grad = None
        tx = x[lo:hi].clone().requires_grad_(True)
        y, pooled = layers.graph_dense_gather(dense, tx)
        ((y * up_y[lo:hi]).sum() + (pooled * up_p[lo:hi]).sum()).backward()
        return y.detach(), pooled.detach(), tx.grad, dense.kernel.grad.clone(), dense.bias.grad.clone()

    y, p, dx, dk, db = run(0, B)
    ak, ab = torch.zeros_like(dk, dtype=torch.float64), torch.zeros_like(db, dtype=torch.float64)
    for lo in range(0, B, CHUNK):
        hi = min(B, lo + CHUNK)
        yc, pc, dc, kc, bc = run(lo, hi)
        assert float((y[lo:hi] - yc).abs().max()) <= 2e-6 * float(y.abs().max())
        assert float((p[lo:hi] - pc).abs().max()) <= 2e-6 * float(p.abs().max())
        assert float((dx[lo:hi] - dc).abs().max()) <= 1e-5 * float(dx.abs().max()), "d inputs of graphs %d..%d" % (lo, hi)
        ak += kc.double(); ab += bc.double()
    assert float((dk.double() - ak).abs().max()) <= 2e-5 * float(ak.abs().max())
    assert float((db.double() - ab).abs().max()) <= 2e-5 * float(ab.abs().max())


@pytest.mark.parametrize("ragged_sizes", [False, True])
def test_graph_batch_normalization_at_600000_rows(ragged_sizes):
    """Learning phase 0 (moving statistics): rows are independent (slices).  Learning phase 1: statistics over the valid rows of the whole batch,
    against the closed form in fp64 on the device (the oracle's formulas, oracle/kgcn_oracle.py: graph_bn_fwd / graph_bn_bwd)."""
    from kgcn_amd import layers
    D = 50
    x, up = _features(B_BIG, D, 12), _features(B_BIG, D, 13)
    g = torch.Generator(device=dev()); g.manual_seed(4)
    sizes = torch.randint(1, N_NODES + 1, (B_BIG,), device=dev(), generator=g) if ragged_sizes else None
    bn = layers.GraphBatchNormalization(activation="sigmoid")
    bn.build((B_BIG, N_NODES, D), dev())
    with torch.no_grad():
        bn.gamma.normal_(1, 0.2); bn.beta.normal_(0, 0.2)
        bn.moving_mean.normal_(0, 0.3); bn.moving_variance.uniform_(0.5, 1.5)
    if not ragged_sizes:
        _check_split(bn, x, up, what="GraphBatchNormalization (learning phase 0)")
    mm, mv = bn.moving_mean.clone(), bn.moving_variance.clone()
    tx = x.clone().requires_grad_(True)
    bn.gamma.grad = bn.beta.grad = None
    bn.learning_phase = 1
    out = bn(tx, enabled_node_nums=sizes)
    out.backward(up)
    valid = torch.ones((B_BIG, N_NODES), dtype=torch.bool, device=dev()) if sizes is None else \
        torch.arange(N_NODES, device=dev())[None, :] < sizes[:, None]
    xd = x.double().requires_grad_(True)
    gam, bet = bn.gamma.detach().double().requires_grad_(True), bn.beta.detach().double().requires_grad_(True)
    rows = xd[valid]
    mean, var = rows.mean(0), rows.var(0, unbiased=False)
    ref = torch.sigmoid((xd - mean) / torch.sqrt(var + bn.eps) * gam + bet) * valid[:, :, None]
    got = out.detach().double() * valid[:, :, None]
    assert float((got - ref.detach()).abs().max()) <= 2e-6
    (ref * up.double()).sum().backward()
    gx = tx.grad.double() * valid[:, :, None]
    assert float((gx - xd.grad * valid[:, :, None]).abs().max()) <= 2e-5 * float(xd.grad.abs().max())
    assert float((bn.gamma.grad.double() - gam.grad).abs().max()) <= 2e-5 * float(gam.grad.abs().max())
    assert float((bn.beta.grad.double() - bet.grad).abs().max()) <= 2e-5 * float(bet.grad.abs().max())
    assert float((bn.moving_mean.double() - (0.99 * mm.double() + 0.01 * mean.detach())).abs().max()) <= 1e-6
    assert float((bn.moving_variance.double() - (0.99 * mv.double() + 0.01 * var.detach())).abs().max()) <= 1e-6


def test_losses_and_adam_at_300000_graphs_and_3m_parameters():
    """models.masked_softmax_ce / masked_sigmoid_ce / sparse_softmax_ce_sum on 300,000 graphs (oracle formulas,
    oracle/kgcn_model_oracle.py:35-61, oracle/kgcn_nets_oracle.py:46-63, evaluated in fp64 on the device) and train.TFAdam on
    3,000,000 parameters in five tensors against the TF update (kgcn/core.py:124) in numpy."""
    from kgcn_amd import models, train
    B = 300_000
    g = torch.Generator(device=dev()); g.manual_seed(6)
    # softmax CE over 2 classes, masked, mean over the padded batch
    lg = (torch.randn((B, 2), device=dev(), generator=g) * 3).requires_grad_(True)
    lab = torch.nn.functional.one_hot(torch.randint(0, 2, (B,), device=dev(), generator=g), 2).float()
    mask = (torch.rand(B, device=dev(), generator=g) < 0.8).float()
    cost, cost_sum = models.masked_softmax_ce(lg, lab, mask)
    cost.backward()
    l64 = lg.detach().double().requires_grad_(True)
    per = -(lab.double() * torch.log_softmax(l64, 1)).sum(1) * mask.double()
    per.mean().backward()
    assert abs(float(cost.detach()) - float(per.mean().detach())) <= 1e-6 * float(per.mean().detach())
    assert abs(float(cost_sum.detach()) - float(per.sum().detach())) <= 1e-6 * float(per.sum().detach())
    assert float((lg.grad.double() - l64.grad).abs().max()) <= 2e-6 * float(l64.grad.abs().max())
    # sigmoid CE over 12 tasks with label mask and pos_weight
    lg = (torch.randn((B, 12), device=dev(), generator=g) * 3).requires_grad_(True)
    lab = (torch.rand((B, 12), device=dev(), generator=g) < 0.3).float()
    ml = (torch.rand((B, 12), device=dev(), generator=g) < 0.8).float()
    pw = 2.5
    cost, cost_sum = models.masked_sigmoid_ce(lg, lab, mask, ml, pw)
    cost.backward()
    l64 = lg.detach().double().requires_grad_(True)
    z = lab.double()
    ce = (1 - z) * l64 + (1 + (pw - 1) * z) * (torch.log1p(torch.exp(-l64.abs())) + torch.clamp(-l64, min=0))
    per = (ce * ml.double()).sum(1) * mask.double()
    per.mean().backward()
    assert abs(float(cost.detach()) - float(per.mean().detach())) <= 1e-6 * float(per.mean().detach())
    assert abs(float(cost_sum.detach()) - float(per.sum().detach())) <= 1e-6 * float(per.sum().detach())
    assert float((lg.grad.double() - l64.grad).abs().max()) <= 2e-6 * float(l64.grad.abs().max())
    # sparse softmax CE, summed (sparse.py:112-113)
    lg = (torch.randn((B, 5), device=dev(), generator=g) * 2).requires_grad_(True)
    idx = torch.randint(0, 5, (B,), device=dev(), generator=g)
    s = models.sparse_softmax_ce_sum(lg, idx)
    s.backward()
    l64 = lg.detach().double().requires_grad_(True)
    r = torch.nn.functional.cross_entropy(l64, idx, reduction="sum")
    r.backward()
    assert abs(float(s.detach()) - float(r.detach())) <= 1e-6 * float(r.detach())
    assert float((lg.grad.double() - l64.grad).abs().max()) <= 2e-6
    # TF-Adam
    shapes = [(1500, 1000), (1000, 1000), (499_000,), (7,), (1000, 1)]
    params = [torch.nn.Parameter(torch.randn(sh, device=dev(), generator=g)) for sh in shapes]
    opt = train.TFAdam(params, lr=1e-2)
    live = list(opt.params)                  # the optimiser re-points the parameters at views of one flat buffer
    p_ref = [p.detach().cpu().numpy().astype(np.float64) for p in live]
    m_ref = [np.zeros_like(p) for p in p_ref]
    v_ref = [np.zeros_like(p) for p in p_ref]
    for t in range(1, 4):
        grads = [torch.randn(p.shape, device=dev(), generator=g) * (0.1 * t) for p in live]
        for p, gr in zip(live, grads):
            p.grad = gr
        opt.step()
        lr_t = 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for i, gr in enumerate(grads):
            g64 = gr.cpu().numpy().astype(np.float64)
            m_ref[i] = 0.9 * m_ref[i] + 0.1 * g64
            v_ref[i] = 0.999 * v_ref[i] + 0.001 * g64 * g64
            p_ref[i] = p_ref[i] - lr_t * m_ref[i] / (np.sqrt(v_ref[i]) + 1e-8)
    for p, r in zip(live, p_ref):
        close(p, r, atol=2e-6, rel=2e-6, what="TF-Adam parameters after 3 steps")
