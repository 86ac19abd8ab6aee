"""The models of BASELINE configs 4 and 5 at the batch sizes bench.py runs them, fused composition against plain
composition of the same HIP layers (each of which is checked against the oracle at sizes the oracle finishes in seconds,
test_gpu_model.py / test_gpu_ragged.py, and layer by layer at full size against oracle/kgcn_ref.c, test_gpu_dense_edges.py):
what only exists at model level AND at this size -- workgroups that walk several tiles with per-tile state (the graph of a
row in the gathered-gradient GEMM), partial sums over > 100,000 rows in the ragged BN / read-out / loss chain -- is what
these catch.  The gathered-gradient GEMM read the pooled gradient of a workgroup's FIRST tile for all its later tiles (rows
beyond 32,768) until such a test existed."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_parity import close, dev  # noqa: E402

pytestmark = pytest.mark.gpu


def _params_like(dst, src):
    with torch.no_grad():
        for (na, a), (nb, b) in zip(dst.named_parameters(), src.named_parameters()):
            assert na == nb and a.shape == b.shape, (na, nb)
            a.copy_(b)


def test_gin_model_at_20000_graphs_fused_read_out_equals_plain_layers():
    """example_model/model_gin.py:40-78 at width 256 on 20,000 ring graphs (200,000 node rows = 3,125 tiles for 512
    workgroups): models.GIN (read-out fused into the block's last GraphDense, d pooled joining inside its dX GEMM, d epsilon
    inside the adjoint aggregation) against GINAggregate / GraphDense / GraphGather called one by one."""
    import bench
    from kgcn_amd import BatchedAdjacency, BatchedCSR, layers, models
    B, N, D = 20_000, 10, 256
    g, r, c, lab, _ = bench.gen_ring_graphs(B, N, seed=5)
    adj = BatchedAdjacency([BatchedCSR.from_arrays(g, r, c, np.ones(g.shape[0], np.float32), B, N, N, device=dev())])
    gen = torch.Generator(device=dev()); gen.manual_seed(5)
    x = torch.randn((B, N, D), device=dev(), generator=gen)
    labels = torch.nn.functional.one_hot(torch.from_numpy(lab), 2).float().to(dev())
    mask = (torch.rand(B, device=dev(), generator=gen) < 0.9).float()
    torch.manual_seed(0)
    model = models.GIN(1, 2, width=D).to(dev())
    model(x, adj)
    with torch.no_grad():
        for a in model.agg:
            a.epsilon[0].fill_(0.3)
        for d in model.dense:
            d.bias.copy_(torch.randn(d.bias.shape, device=dev(), generator=gen) * 0.1)

    def plain(feat):
        layer, outs = feat, []
        for blk in range(2):
            layer = model.agg[blk](layer, adj=adj)
            layer = model.dense[2 * blk](layer)
            layer = model.dense[2 * blk + 1](layer)
            outs.append(layers.GraphGather()(layer))
        return model.out(torch.cat(outs, dim=1))

    res = []
    for fn in (lambda f: model(f, adj), plain):
        model.zero_grad(set_to_none=True)
        tx = x.clone().requires_grad_(True)
        logits = fn(tx)
        cost, _ = models.masked_softmax_ce(logits, labels, mask)
        cost.backward()
        res.append((logits.detach(), float(cost.detach()), tx.grad, [(n_, p.grad.clone()) for n_, p in model.named_parameters()]))
    (la, ca, xa, pa), (lb, cb, xb, pb) = res
    scale = float(lb.abs().max())
    close(la, lb.cpu().numpy(), atol=2e-6 * scale, rel=2e-6, what="GIN logits, fused vs plain")
    assert abs(ca - cb) <= 2e-6 * abs(cb)
    close(xa, xb.cpu().numpy(), atol=1e-6 * float(xb.abs().max()), rel=2e-5, what="GIN d features")
    for (n_, a), (_, b) in zip(pa, pb):
        close(a, b.cpu().numpy(), atol=2e-6 * float(b.abs().max()), rel=2e-5, what="GIN grad %s" % n_)


def test_multitask_model_at_batch_4096_ragged_equals_padded():
    """example_model/model_multitask.py:45-101 on 4,096 Tox21-shaped molecules (N = 50 padded, true sizes 5..50: 204,800 padded /
    ~113,000 valid node rows): the valid-rows-only execution bench.py --config cfg4 times (kgcn_amd.ragged: aggregate-first first
    layer, row-chunk aggregation, BN statistics, the padding rows' closed-form share of the read-out) against the padded
    formulation of the reference on the same HIP layers: logits, loss, every parameter gradient."""
    import bench
    from kgcn_amd import BatchedAdjacency, BatchedCSR, data_util as D, models
    B, N, F, TASKS = 4096, 50, 81, 12
    sizes, g, r, c, rng = bench.gen_tox21_like(B, N, seed=4)
    chan = D.normalize_adj(D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), B, N))
    adj = BatchedAdjacency([BatchedCSR.from_arrays(chan.graph, chan.row, chan.col, chan.val, B, N, N, device=dev())])
    valid = np.arange(N)[None, :] < sizes[:, None]
    x = torch.from_numpy(rng.standard_normal((B, N, F)).astype(np.float32) * valid[:, :, None]).to(dev())
    labels = torch.from_numpy((rng.random((B, TASKS)) < 0.3).astype(np.float32)).to(dev())
    mask_label = torch.from_numpy((rng.random((B, TASKS)) < 0.8).astype(np.float32)).to(dev())
    mask = torch.from_numpy((rng.random(B) < 0.95).astype(np.float32)).to(dev())
    en = torch.from_numpy(sizes.astype(np.int32)).to(dev())
    res = []
    ref_model = None
    for ragged in (True, False):
        torch.manual_seed(0)
        model = models.MultitaskGCN(1, TASKS, ragged=ragged).to(dev())
        model(x, adj, enabled_node_nums=en)
        if ref_model is None:
            gen = torch.Generator(device="cpu").manual_seed(1)
            with torch.no_grad():
                for p in model.parameters():
                    if p.dim() == 1 or p.shape[0] == 1:
                        p.add_(torch.randn(p.shape, generator=gen).to(p.device) * 0.1)
            ref_model = model
        else:
            _params_like(model, ref_model)
        logits = model(x, adj, enabled_node_nums=en)
        cost, cost_sum = models.masked_sigmoid_ce(logits, labels, mask, mask_label, 2.0)
        cost.backward()
        res.append((logits.detach(), float(cost.detach()), float(cost_sum.detach()), [(n_, p.grad.clone()) for n_, p in model.named_parameters()]))
    (la, ca, sa, pa), (lb, cb, sb_, pb) = res
    close(la, lb.cpu().numpy(), atol=2e-5, rel=2e-5, what="multitask logits, ragged vs padded")
    assert abs(ca - cb) <= 1e-5 * abs(cb) and abs(sa - sb_) <= 1e-5 * abs(sb_)
    for (n_, a), (_, b) in zip(pa, pb):
        close(a, b.cpu().numpy(), atol=2e-5 * float(b.abs().max()), rel=5e-5, what="multitask grad %s" % n_)
