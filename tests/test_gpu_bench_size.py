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


# ---------------------------------------------------------------------------------------------------------------------
# The same two models at the same sizes against an INDEPENDENT oracle chain (fp64 numpy on block-diagonal CSR:
# oracle/kgcn_nets_oracle.py, itself checked by finite differences in tests/test_oracle_model.py): logits, loss and every
# parameter gradient of what `bench.py --config cfg4 | cfg5` times.
# ---------------------------------------------------------------------------------------------------------------------
def _coo_lists(g, r, c, val, B, N):
    order = np.argsort(g, kind="stable")
    g, r, c, val = g[order], r[order], c[order], val[order]
    off = np.zeros(B + 1, np.int64)
    np.cumsum(np.bincount(g, minlength=B), out=off[1:])
    idx = np.stack([r, c], 1).astype(np.int32)
    return [[(idx[off[b]:off[b + 1]], val[off[b]:off[b + 1]], [N, N])] for b in range(B)]


def test_cfg4_model_at_batch_4096_against_the_oracle():
    """example_model/model_multitask.py:45-101 at 4,096 Tox21-shaped molecules on the valid rows only (what bench.py --config
    cfg4 runs) against the fp64 PADDED formulation of the oracle."""
    import bench
    from oracle import kgcn_nets_oracle as NETS
    from kgcn_amd import BatchedAdjacency, BatchedCSR, data_util as D, models
    B, N, F, TASKS = 4096, 50, 81, 12
    sizes, g, r, c, rng = bench.gen_tox21_like(B, N, seed=4)
    chan = D.normalize_adj(D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), B, N))
    adj = BatchedAdjacency([BatchedCSR.from_arrays(chan.graph, chan.row, chan.col, chan.val, B, N, N, device=dev())])
    adjs = _coo_lists(np.asarray(chan.graph), np.asarray(chan.row), np.asarray(chan.col), np.asarray(chan.val), B, N)
    valid = np.arange(N)[None, :] < sizes[:, None]
    x = (rng.standard_normal((B, N, F)).astype(np.float32) * valid[:, :, None]).astype(np.float32)
    labels = (rng.random((B, TASKS)) < 0.3).astype(np.float32)
    mask_label = (rng.random((B, TASKS)) < 0.8).astype(np.float32)
    mask = (rng.random(B) < 0.95).astype(np.float32)
    pos_weight = (mask_label.sum(0) - (labels * mask_label).sum(0) + 0.01) / ((labels * mask_label).sum(0) + 0.01)     # per task
    p = NETS.multitask_init(np.random.default_rng(8), F, TASKS)
    for k in ("b1", "b2", "b4"):
        p[k] = [np.random.default_rng(9).standard_normal(p[k][0].shape) * 0.1]
    p["c5"] = np.random.default_rng(10).standard_normal(p["c5"].shape) * 0.1
    x64 = x.astype(np.float64)
    cc = NETS.multitask_forward(p, x64, adjs, labels.astype(np.float64), mask.astype(np.float64), mask_label.astype(np.float64), sizes, pos_weight)
    gg = NETS.multitask_backward(p, cc, x64, adjs, labels.astype(np.float64), mask.astype(np.float64), mask_label.astype(np.float64), pos_weight)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev())
    model = models.MultitaskGCN(1, TASKS, ragged=True).to(dev())
    tx, en = t(x), torch.from_numpy(sizes.astype(np.int32)).to(dev())
    model(tx, adj, enabled_node_nums=en)
    with torch.no_grad():
        for conv, w, b in ((model.conv1, "w1", "b1"), (model.conv2, "w2", "b2"), (model.conv3, "w4", "b4")):
            conv.w[0].copy_(t(p[w][0])); conv.bias[0].copy_(t(p[b][0]))
        model.dense1.kernel.copy_(t(p["k3"])); model.dense1.bias.copy_(t(p["c3"]))
        model.dense2.kernel.copy_(t(p["k5"])); model.dense2.bias.copy_(t(p["c5"]))
        model.out.kernel.copy_(t(p["ok"])); model.out.bias.copy_(t(p["ob"]))
    logits = model(tx, adj, enabled_node_nums=en)
    close(logits, cc["logits"], atol=1e-5, rel=1e-5, what="cfg4 @4096 logits vs oracle")
    cost_opt, cost_sum = models.masked_sigmoid_ce(logits, t(labels), t(mask), t(mask_label), pos_weight)
    assert abs(float(cost_opt) - cc["cost_opt"]) < 1e-5 * abs(cc["cost_opt"]) and abs(float(cost_sum) - cc["cost_sum"]) < 1e-5 * abs(cc["cost_sum"])
    cost_opt.backward()
    for name, tt, ref in [("w1", model.conv1.w[0], gg["w1"][0]), ("b1", model.conv1.bias[0], gg["b1"][0]),
                          ("w2", model.conv2.w[0], gg["w2"][0]), ("b2", model.conv2.bias[0], gg["b2"][0]),
                          ("k3", model.dense1.kernel, gg["k3"]), ("c3", model.dense1.bias, gg["c3"]),
                          ("w4", model.conv3.w[0], gg["w4"][0]), ("b4", model.conv3.bias[0], gg["b4"][0]),
                          ("gamma", model.bn.gamma, gg["gamma"]), ("beta", model.bn.beta, gg["beta"]),
                          ("k5", model.dense2.kernel, gg["k5"]), ("c5", model.dense2.bias, gg["c5"]),
                          ("ok", model.out.kernel, gg["ok"]), ("ob", model.out.bias, gg["ob"])]:
        close(tt.grad, np.asarray(ref).reshape(tuple(tt.shape)), atol=0, rel=1e-5, what="cfg4 @4096 grad %s vs oracle" % name)


def test_cfg5_model_at_20000_graphs_against_the_oracle():
    """example_model/model_gin.py:40-78 at width 256 on 20,000 ring graphs (what bench.py --config cfg5 runs) against
    oracle/kgcn_nets_oracle.gin_*: logits, loss, every parameter gradient, d features."""
    import bench
    from oracle import kgcn_nets_oracle as NETS
    from kgcn_amd import BatchedAdjacency, BatchedCSR, models
    B, N, Dm = 20_000, 10, 256
    g, r, c, lab, _ = bench.gen_ring_graphs(B, N, seed=5)
    ones = np.ones(g.shape[0], np.float32)
    adj = BatchedAdjacency([BatchedCSR.from_arrays(g, r, c, ones, B, N, N, device=dev())])
    adjs = _coo_lists(np.asarray(g), np.asarray(r), np.asarray(c), ones, B, N)
    rng = np.random.default_rng(15)
    x = rng.standard_normal((B, N, Dm)).astype(np.float32)
    labels = np.eye(2, dtype=np.float32)[lab]
    mask = (rng.random(B) < 0.9).astype(np.float32)
    p = NETS.gin_init(np.random.default_rng(16), Dm, Dm)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=dev())
    model = models.GIN(1, 2, width=Dm).to(dev())
    tx = t(x).requires_grad_(True)
    model(tx.detach(), adj)
    with torch.no_grad():
        for blk in range(2):
            model.agg[blk].epsilon[0].fill_(float(p["eps"][blk][0]))
        for i in range(4):
            model.dense[i].kernel.copy_(t(p["k%d" % i])); model.dense[i].bias.copy_(t(p["c%d" % i]))
        model.out.kernel.copy_(t(p["ok"])); model.out.bias.copy_(t(p["ob"]))
        for blk in range(2):        # the oracle computes with the float32 values the device holds
            p["eps"][blk][0] = float(model.agg[blk].epsilon[0])
    x64 = x.astype(np.float64)
    cc = NETS.gin_forward(p, x64, adjs, labels.astype(np.float64), mask.astype(np.float64))
    logits = model(tx, adj)
    close(logits, cc["logits"], atol=0, rel=1e-5, what="cfg5 @20000 logits vs oracle")
    cost_opt, cost_sum = models.masked_softmax_ce(logits, t(labels), t(mask))
    assert abs(float(cost_opt) - cc["cost_opt"]) < 1e-5 * abs(cc["cost_opt"])
    cost_opt.backward()
    # relu masks from the DEVICE activations: of 51 M pre-activations per layer a dozen lie within fp32 rounding of zero, and ONE
    # flipped mask moves a kernel-gradient column by |input row| x |gradient| ~ 1e-5 -- 3e-4 of the largest entry (measured)
    masks = {}
    with torch.no_grad():
        h = tx.detach()
        for blk in range(2):
            h = model.agg[blk](h, adj=adj)
            h = model.dense[2 * blk](h); masks[(blk, 0)] = (h > 0).cpu().numpy()
            h = model.dense[2 * blk + 1](h); masks[(blk, 1)] = (h > 0).cpu().numpy()
    # ... and that substitution is COUNTED, not assumed: a device mask may differ from the fp64 oracle's own only where the
    # pre-activation lies within fp32 rounding of zero -- at most 64 of the 51 M entries of a layer, each with an oracle
    # pre-activation below 1e-5 of the layer's largest ("a dozen" was a comment, now it is a check)
    for (blk, lay), mk in masks.items():
        z = cc["z%d_%d" % (lay, blk)]
        differ = mk.reshape(z.shape) != (z > 0)
        n_diff = int(differ.sum())
        worst = float(np.abs(z[differ]).max()) if n_diff else 0.0
        assert n_diff <= 64, "block %d layer %d: %d of %d relu masks differ from the fp64 oracle's" % (blk, lay, n_diff, z.size)
        assert worst <= 1e-5 * float(np.abs(z).max()), (blk, lay, n_diff, worst)
    gg = NETS.gin_backward(p, cc, x64, adjs, labels.astype(np.float64), mask.astype(np.float64), relu_masks=masks)
    for i in range(4):
        close(model.dense[i].kernel.grad, gg["k%d" % i], atol=0, rel=2e-5, what="cfg5 @20000 grad dense%d kernel vs oracle" % i)
        close(model.dense[i].bias.grad, gg["c%d" % i], atol=0, rel=2e-5, what="cfg5 @20000 grad dense%d bias vs oracle" % i)
    close(model.out.kernel.grad, gg["ok"], atol=0, rel=1e-5, what="cfg5 @20000 grad out kernel vs oracle")
    close(model.out.bias.grad, gg["ob"], atol=0, rel=1e-5, what="cfg5 @20000 grad out bias vs oracle")
    close(tx.grad, gg["dx"], atol=0, rel=2e-5, what="cfg5 @20000 d features vs oracle")
    for blk in range(2):            # d eps = <d aggregate, input>: a signed sum of 51 M products; tolerance relative to sum |terms|
        terms = float(np.abs(cc["in%d" % blk]).mean()) * 1e-6
        assert abs(float(model.agg[blk].epsilon[0].grad) - gg["eps"][blk][0]) <= 2e-5 * max(abs(gg["eps"][blk][0]), terms), blk
