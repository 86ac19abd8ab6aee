"""Model-level parity: example_model/model.py's network on synthetic.jbl batches,
HIP path (fp32) vs the fp64 model oracle -- logits, loss, every parameter gradient, and the
parameter trajectory over several TF-Adam steps; plus a short training run (loss must fall) through
the product's own loaders for GCN and GIN."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_model_oracle as M
from test_gpu_parity import close, dev, t32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _set_params(model, p):
    with torch.no_grad():
        for i, conv in enumerate((model.conv1, model.conv2, model.conv3), 1):
            conv.w[0].copy_(t32(p["w%d" % i][0]))
            conv.bias[0].copy_(t32(p["b%d" % i][0]))
        model.bn.gamma.copy_(t32(p["gamma"])); model.bn.beta.copy_(t32(p["beta"]))
        model.dense.kernel.copy_(t32(p["dk"])); model.dense.bias.copy_(t32(p["db"]))
        model.out.kernel.copy_(t32(p["ok"])); model.out.bias.copy_(t32(p["ob"]))


def _named(model):
    return {"w1": model.conv1.w[0], "b1": model.conv1.bias[0], "w2": model.conv2.w[0], "b2": model.conv2.bias[0],
            "w3": model.conv3.w[0], "b3": model.conv3.bias[0], "gamma": model.bn.gamma, "beta": model.bn.beta,
            "dk": model.dense.kernel, "db": model.dense.bias, "ok": model.out.kernel, "ob": model.out.bias}


@pytest.mark.parametrize("batch", ["g3_synthetic_feed_full30.npz", "g3_synthetic_feed_b30.npz"])
def test_model_py_gradients_and_adam_trajectory(batch):
    from kgcn_amd import models, train
    z = load_golden(batch)
    adjs = unflatten_adjs(z, "adj_")
    x, labels, mask = z["features"], z["labels"].astype(np.float64), z["mask"].astype(np.float64)
    p = M.init_params(np.random.default_rng(5), 3)
    model = models.GCN(1).to(dev())
    tx, tl, tm = t32(x), t32(labels), t32(mask)
    model(tx, adjs)                                   # builds the lazily created parameters
    _set_params(model, p)
    opt = train.TFAdam(model.parameters(), lr=0.01)
    oopt = M.TFAdam(lr=0.01)
    xo = x.astype(np.float64)
    for step in range(6):
        c = M.forward(p, xo, adjs, labels, mask)
        g = M.backward(p, c, xo, adjs, labels, mask)
        opt.zero_grad()
        logits = model(tx, adjs)
        cost_opt, cost_sum = models.masked_softmax_ce(logits, tl, tm)
        cost_opt.backward()
        if step == 0:
            close(logits, c["logits"], atol=1e-5, what="logits")      # measured 1.1e-6 (profiles/r04_accuracy.json)
            assert abs(float(cost_opt) - c["cost_opt"]) < 5e-6 and abs(float(cost_sum) - c["cost_sum"]) < 1e-4
            for k, t in _named(model).items():
                ref = g[k][0] if isinstance(g[k], list) else g[k]
                close(t.grad, np.asarray(ref).reshape(tuple(t.shape)), atol=1e-6, rel=1e-5, what="grad " + k)
        opt.step()
        p = oopt.step(p, g)
    for k, t in _named(model).items():               # parameters after 6 Adam steps
        ref = p[k][0] if isinstance(p[k], list) else p[k]
        close(t, np.asarray(ref).reshape(tuple(t.shape)), atol=2e-6, what="param " + k)    # measured <= 1.9e-7 (profiles/r05_accuracy.json): a misplaced Adam epsilon moves a parameter by ~1e-4


@pytest.mark.parametrize("kind", ["GCN", "GIN", "GAT"])
def test_short_training_run_on_synthetic_jbl(kind):
    """kgcn train --config example_config/synth.json in miniature: batch 30, padded last batch,
    shuffled epochs, TF Adam; through the product's loaders.  The loss must fall."""
    from kgcn_amd import data_util as D, models, train
    raw = load_golden("g1_synthetic_raw.npz")
    chans, enabled = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
    feats, labels = raw["feature"], raw["label"]
    model = {"GCN": models.GCN, "GIN": models.GIN, "GAT": models.GATNet}[kind](1).to(dev())
    rng = np.random.default_rng(1234)
    idx = np.arange(160)
    opt = None
    epoch_cost = []
    for epoch in range(8):
        rng.shuffle(idx)
        tot = 0.0
        for it in range(6):
            bidx = idx[it * 30:(it + 1) * 30]
            adj = D.batch_adjacency(chans, bidx, 30, device=dev())
            x = D.batch_features(feats, bidx, 30, device=dev())
            lab = torch.zeros((30, 2), device=dev()); lab[:len(bidx)] = t32(labels[bidx])
            mask = torch.zeros(30, device=dev()); mask[:len(bidx)] = 1
            if opt is None:
                model(x, adj)
                opt = train.TFAdam(model.parameters(), lr=0.01)
            cs, _ = train.train_step(model, opt, models.masked_softmax_ce, x, adj, lab, mask)
            tot += cs
        epoch_cost.append(tot / 160)
    assert np.isfinite(epoch_cost).all() and epoch_cost[-1] < epoch_cost[0] - 0.02, epoch_cost


# ---- example_model/model_multitask.py (BASELINE config 4) and example_model/sparse.py (config 3) ------
from oracle import kgcn_nets_oracle as NETS
from oracle import kgcn_oracle as K


def _copy_conv(conv, w, b):
    with torch.no_grad():
        for c in range(len(w)):
            conv.w[c].copy_(t32(w[c])); conv.bias[c].copy_(t32(b[c]))


def _grad_of(t, ref, what, rel=2e-5):
    close(t.grad, np.asarray(ref).reshape(tuple(t.shape)), atol=2e-6, rel=rel, what="grad " + what)


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("pos_weight", [None, 2.5, "per_task"])
def test_model_multitask_tox21_shaped(pos_weight, ragged):
    """12 tasks, N = 50 padded with variable true sizes, F = 81, widths 256/256/256/50/50, masked labels,
    a dummy graph in the batch; logits, loss and every gradient vs the fp64 oracle (the PADDED formulation of the
    reference).  ragged=True: the product computes on the valid node rows only (kgcn_amd.ragged) -- same numbers, the padded
    rows' constant share of GraphGather and its gradient into dense2 / bias included."""
    from kgcn_amd import models
    from test_oracle_model import tox21_like_batch
    rng = np.random.default_rng(44)
    x, adjs, labels, mask, mask_label, sizes = tox21_like_batch(rng, B=24, N=50, F=81, T=12)
    if isinstance(pos_weight, str):          # the reference's info.pos_weight: one weight per label column (kgcn/data_util.py:563-568)
        pos_weight = (np.nansum(mask_label, 0) - np.nansum(labels * mask_label, 0) + 0.01) / (np.nansum(labels * mask_label, 0) + 0.01)
        assert pos_weight.shape == (12,) and len(set(np.round(pos_weight, 3))) > 3
    p = NETS.multitask_init(rng, 81, 12)
    for k in ("b1", "b2", "b4"):
        p[k] = [rng.standard_normal(p[k][0].shape) * 0.1]
    c = NETS.multitask_forward(p, x, adjs, labels, mask, mask_label, sizes, pos_weight)
    g = NETS.multitask_backward(p, c, x, adjs, labels, mask, mask_label, pos_weight)
    p["c5"] = rng.standard_normal(p["c5"].shape) * 0.1                # a bias the padded rows' constant row depends on
    c = NETS.multitask_forward(p, x, adjs, labels, mask, mask_label, sizes, pos_weight)
    g = NETS.multitask_backward(p, c, x, adjs, labels, mask, mask_label, pos_weight)
    model = models.MultitaskGCN(1, 12, ragged=ragged).to(dev())
    tx = t32(x).requires_grad_(True)
    en = torch.as_tensor(sizes)
    model(tx, adjs, enabled_node_nums=en)
    _copy_conv(model.conv1, p["w1"], p["b1"]); _copy_conv(model.conv2, p["w2"], p["b2"])
    _copy_conv(model.conv3, p["w4"], p["b4"])
    with torch.no_grad():
        model.dense1.kernel.copy_(t32(p["k3"])); model.dense1.bias.copy_(t32(p["c3"]))
        model.dense2.kernel.copy_(t32(p["k5"])); model.dense2.bias.copy_(t32(p["c5"]))
        model.out.kernel.copy_(t32(p["ok"])); model.out.bias.copy_(t32(p["ob"]))
    logits = model(tx, adjs, enabled_node_nums=en)
    # logits reach 61: one fp32 ulp there is 3.8e-6 and six 256-wide layers sit in front of them -- the bound is RELATIVE to the
    # largest logit, 1e-6 (measured 5.2e-7 padded, 1.6e-7 ragged; profiles/r04_accuracy.json)
    close(logits, c["logits"], atol=1e-6 * max(1.0, float(np.abs(c["logits"]).max())), what="multitask logits")
    cost_opt, cost_sum = models.masked_sigmoid_ce(logits, t32(labels), t32(mask), t32(mask_label), pos_weight)
    assert abs(float(cost_opt) - c["cost_opt"]) < 1e-5 * max(1.0, abs(c["cost_opt"]))
    assert abs(float(cost_sum) - c["cost_sum"]) < 1e-5 * max(1.0, abs(c["cost_sum"]))
    cost_opt.backward()
    _grad_of(tx, g["dx"], "x")
    for name, t, ref in [("w1", model.conv1.w[0], g["w1"][0]), ("b1", model.conv1.bias[0], g["b1"][0]),
                         ("w2", model.conv2.w[0], g["w2"][0]), ("k3", model.dense1.kernel, g["k3"]),
                         ("c3", model.dense1.bias, g["c3"]), ("w4", model.conv3.w[0], g["w4"][0]),
                         ("b4", model.conv3.bias[0], g["b4"][0]), ("gamma", model.bn.gamma, g["gamma"]),
                         ("beta", model.bn.beta, g["beta"]), ("k5", model.dense2.kernel, g["k5"]),
                         ("ok", model.out.kernel, g["ok"]), ("ob", model.out.bias, g["ob"])]:
        _grad_of(t, ref, name)


@pytest.mark.parametrize("mode", ["normalize", "split"])
def test_model_sparse_block_diagonal(mode):
    """kgcn-sparse: records -> product block-diagonal builder -> SparseGCN (batch of ONE [sumN x sumN] graph,
    1 channel Kipf-normalised or max_degree+1 = 6 split channels) vs the oracle built from the
    per-molecule restatement; relu masks of the oracle chain rule taken from the GPU activations."""
    from kgcn_amd import data_util as D, models
    from test_oracle_model import _construct, _sparse_batch
    rng = np.random.default_rng(33)
    F, ncls = 40, 5
    f, sizes = _sparse_batch(rng, nmol=24, F=F)
    kw = dict(max_degree=0, normalize=True) if mode == "normalize" else dict(max_degree=5, normalize=False, split_adj=True)
    chans, net = _construct(f, F, **kw)
    batch = D.block_diagonal_batch(f["size"][:, 0], f["adj_row"], f["adj_column"], f["adj_values"], f["adj_elem_len"],
                                   f["adj_degrees"], f["feature_row"], f["feature_column"], f["feature_values"],
                                   f["feature_elem_len"], F, device=dev(), **kw)
    labels = rng.integers(0, ncls, size=len(sizes))
    C = len(chans)
    p = NETS.sparse_init(rng, F, ncls, channels=C, out_dims=(256, 256, 256), dense_dim=256)
    for i in (1, 2, 3):
        p["b%d" % i] = [rng.standard_normal(b.shape) * 0.05 for b in p["b%d" % i]]
    model = models.SparseGCN(ncls, adj_channel_num=C).to(dev())
    model(batch)
    for i, conv in enumerate(model.convs, 1):
        _copy_conv(conv, p["w%d" % i], p["b%d" % i])
    with torch.no_grad():
        model.dense.kernel.copy_(t32(p["dk"])); model.dense.bias.copy_(t32(p["dc"]))
        model.out.kernel.copy_(t32(p["ok"])); model.out.bias.copy_(t32(p["ob"]))
    logits = model(batch)
    c = NETS.sparse_forward(p, net, chans, sizes, labels)
    close(logits, c["logits"], atol=1e-5, what="sparse logits")      # measured 8.4e-7
    loss = models.sparse_softmax_ce_sum(logits, torch.as_tensor(labels, device=dev()))
    assert abs(float(loss) - c["loss"]) < 1e-5 * max(1.0, abs(c["loss"]))
    loss.backward()
    g = NETS.sparse_backward(p, c, chans, sizes, labels)
    for i, conv in enumerate(model.convs, 1):
        for ch in range(C):
            _grad_of(conv.w[ch], g["w%d" % i][ch], "w%d[%d]" % (i, ch), rel=1e-5)       # measured <= 1.05e-6 of max(1, |ref|)
            _grad_of(conv.bias[ch], g["b%d" % i][ch], "b%d[%d]" % (i, ch), rel=1e-5)
    _grad_of(model.dense.kernel, g["dk"], "dk", rel=1e-5); _grad_of(model.out.kernel, g["ok"], "ok", rel=1e-5)
    _grad_of(model.bn.gamma, g["gamma"], "gamma", rel=1e-5); _grad_of(model.bn.beta, g["beta"], "beta", rel=1e-5)


def test_graphed_train_step_matches_eager():
    """The hipGraph-captured train step (static batch buffers refilled on the device, capturable TF-Adam)
    follows the eager step: same costs and the same parameters after two epochs with a padded last batch."""
    from kgcn_amd import data_util as D, models, train
    raw = load_golden("g1_synthetic_raw.npz")
    chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
    feats, labels = raw["feature"], raw["label"]
    ds = D.DeviceGraphDataset(chans, feats, device=dev())
    torch.manual_seed(0)
    m_e, m_g = models.GCN(1).to(dev()), models.GCN(1).to(dev())
    adj0, x0 = ds.batch(np.arange(30), 30)
    m_e(x0, adj0); m_g(x0, adj0)                                     # lazy parameter creation
    m_g.load_state_dict(m_e.state_dict())
    o_e = train.TFAdam(m_e.parameters(), lr=0.01)
    o_g = train.TFAdam(m_g.parameters(), lr=0.01, capturable=True)
    sb = ds.static_batch(30)
    lab = torch.zeros((30, 2), device=dev())
    mask = torch.zeros(30, device=dev())
    sb.load(np.arange(30))
    step = train.GraphedTrainStep(m_g, o_g, models.masked_softmax_ce, sb, lab, mask)
    for a, b in zip(m_e.parameters(), m_g.parameters()):            # capture did not train
        assert torch.equal(a, b)
    rng = np.random.default_rng(2)
    idx = np.arange(160)
    for epoch in range(2):
        rng.shuffle(idx)
        for it in range(6):
            bidx = idx[it * 30:(it + 1) * 30]
            nb = len(bidx)
            lab.zero_(); lab[:nb] = t32(labels[bidx])
            mask.zero_(); mask[:nb] = 1
            adj, x = ds.batch(bidx, 30)
            cs_e, lg_e = train.train_step(m_e, o_e, models.masked_softmax_ce, x, adj, lab, mask)
            sb.load(bidx)
            cs_g, lg_g = step.replay()
            assert abs(cs_e - float(cs_g)) < 1e-4 * max(1.0, abs(cs_e)), (epoch, it, cs_e, float(cs_g))
            close(lg_g, lg_e.cpu().numpy(), atol=1e-6, what="logits")                # measured 0.0: the replay launches the same kernels
    assert o_g.t == o_e.t == 12 and float(o_g._t_dev) == 12
    for a, b in zip(m_e.parameters(), m_g.parameters()):
        close(b, a.detach().cpu().numpy(), atol=1e-7, what="params after 12 steps")


@pytest.mark.parametrize("kind", ["GIN", "GAT", "GCN-split"])
def test_graphed_train_step_other_models(kind):
    """Every kernel family of the path inside a hipGraph capture (GIN aggregate + d eps dot, GAT, the unfused
    multi-channel GraphConv route): replaying the captured step trains -- the loss of the same batch falls and the
    replay matches an eager step taken from the same state."""
    from kgcn_amd import data_util as D, models, train
    raw = load_golden("g1_synthetic_raw.npz")
    chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10},
                            split_adj_flag=(kind == "GCN-split"))
    ds = D.DeviceGraphDataset(chans, raw["feature"], device=dev())
    C = len(chans)
    make = {"GIN": models.GIN, "GAT": models.GATNet, "GCN-split": models.GCN}[kind]
    torch.manual_seed(1)
    m_g, m_e = make(C).to(dev()), make(C).to(dev())
    idx = np.arange(30)
    adj0, x0 = ds.batch(idx, 30)
    m_g(x0, adj0); m_e(x0, adj0)
    m_e.load_state_dict(m_g.state_dict())
    lab = t32(raw["label"][idx].astype(np.float32))
    mask = torch.ones(30, device=dev())
    o_g = train.TFAdam(m_g.parameters(), lr=0.01, capturable=True)
    o_e = train.TFAdam(m_e.parameters(), lr=0.01)
    sb = ds.static_batch(30)
    sb.load(idx)
    step = train.GraphedTrainStep(m_g, o_g, models.masked_softmax_ce, sb, lab, mask)
    costs = []
    for it in range(5):
        cs_e, _ = train.train_step(m_e, o_e, models.masked_softmax_ce, x0, adj0, lab, mask)
        cs_g, _ = step.replay()
        costs.append(float(cs_g))
        assert abs(cs_e - costs[-1]) < 2e-4 * max(1.0, abs(cs_e)), (kind, it, cs_e, costs[-1])
    assert costs[-1] < costs[0]


@pytest.mark.parametrize("kind", ["GCN", "GCN-split", "GIN", "multitask-ragged"])
def test_deferred_second_stages_leave_every_gradient_unchanged(kind):
    """ops.deferred_reductions (one second-stage launch per training step) must be invisible: every parameter gradient of a
    backward pass inside the block is BIT-equal to the one taken without deferral -- also where a gradient is read inside the
    pass (batch normalisation's d gamma / d beta enter its dx; the concatenated kernels of a multi-channel GraphConv are sliced
    by autograd; the ragged multitask model uses its last dense layer twice)."""
    from kgcn_amd import data_util as D, models, ops
    torch.manual_seed(3)
    if kind == "multitask-ragged":
        from test_oracle_model import tox21_like_batch
        x, adj0, labels, mask, mask_label, sizes = tox21_like_batch(np.random.default_rng(5), B=24, N=50, F=81, T=12)
        model = models.MultitaskGCN(1, 12, ragged=True).to(dev())
        x0, kw = t32(x), {"enabled_node_nums": torch.as_tensor(sizes)}
        loss = lambda lg: models.masked_sigmoid_ce(lg, t32(labels), t32(mask), t32(mask_label), 1.0)[0]
    else:
        raw = load_golden("g1_synthetic_raw.npz")
        chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10},
                                split_adj_flag=(kind == "GCN-split"))
        ds = D.DeviceGraphDataset(chans, raw["feature"], device=dev())
        model = {"GCN": models.GCN, "GCN-split": models.GCN, "GIN": models.GIN}[kind](len(chans)).to(dev())
        lab = t32(raw["label"][:30].astype(np.float32))
        mask = torch.ones(30, device=dev())
        loss = lambda lg: models.masked_softmax_ce(lg, lab, mask)[0]
        adj0, x0 = ds.batch(np.arange(30), 30)
        kw = {}

    def grads(defer):
        for p in model.parameters():
            p.grad = None
        ops.weight_tables.refresh()
        cost = loss(model(x0, adj0, **kw))
        if defer:
            with ops.deferred_reductions():
                cost.backward()
        else:
            cost.backward()
        torch.cuda.synchronize()
        return [(n, p.grad.clone()) for n, p in model.named_parameters()]

    grads(False)
    for trial in range(3):
        plain, waited = grads(False), grads(True)
        for (n, a), (_, b) in zip(plain, waited):
            assert torch.equal(a, b), (kind, trial, n, float((a - b).abs().max()))


def _grads_with_and_without_deferral(params, make_loss):
    from kgcn_amd import ops

    def grads(defer):
        for p in params:
            p.grad = None
        ops.weight_tables.refresh()
        cost = make_loss()
        if defer:
            with ops.deferred_reductions(root=cost):
                cost.backward()
        else:
            cost.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    grads(False)
    for trial in range(3):
        plain, waited = grads(False), grads(True)
        for i, (a, b) in enumerate(zip(plain, waited)):
            assert torch.equal(a, b), (trial, i, float((a - b).abs().max()))
    return plain


def test_deferral_sees_a_graphconv_applied_twice_on_the_aggregate_first_route():
    """stack_rows' [w; bias; pad] is a fresh tensor on every call, so counting ITS uses never saw that a
    GraphConv on the aggregate-first route (din + 1 < dout, C == 1, >= 1,024 rows) applied twice in one step shares w / bias:
    both weight gradients waited for the flush while autograd added the two views inside the pass.  The uses are counted on the
    parameters behind the operand now, and the block checks the autograd graph (two edges into w's AccumulateGrad)."""
    from kgcn_amd import layers
    from oracle import kgcn_oracle as K
    rng = np.random.default_rng(11)
    T, N, F, H = 64, 20, 12, 32                      # 1,280 rows: aggregate-first (dp = 16 < 32)
    adjs = K.synth_mol_graphs(rng, T, N, 2)
    x = t32(rng.standard_normal((T, N, F)).astype(np.float32))
    conv = layers.GraphConv(H, 1).to(dev())
    lift = torch.nn.Parameter(t32(rng.standard_normal((H, F)).astype(np.float32) * 0.2))
    g = t32(rng.standard_normal((T, N, H)).astype(np.float32))
    conv(x, adj=adjs)                                 # build
    params = [conv.w[0], conv.bias[0], lift]

    def make_loss():
        h = torch.tanh(conv(x, adj=adjs))
        h2 = conv(h @ lift, adj=adjs)                 # the SAME layer again (weight sharing)
        return (h2 * g).sum()

    plain = _grads_with_and_without_deferral(params, make_loss)
    # and the shared gradient is the sum of both uses: against the oracle
    w, b = conv.w[0].detach().cpu().numpy().astype(np.float64), conv.bias[0].detach().cpu().numpy().astype(np.float64)
    xn, ln, gn = x.cpu().numpy().astype(np.float64), lift.detach().cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)
    o1 = K.graphconv_fwd(xn, adjs, [w], [b])
    h = np.tanh(o1)
    dx2, dw2, db2 = K.graphconv_bwd(h @ ln, adjs, [w], [b], gn)
    dh = dx2 @ ln.T
    _, dw1, db1 = K.graphconv_bwd(xn, adjs, [w], [b], dh * (1.0 - h * h))
    close(plain[0], dw1[0] + dw2[0], 0.0, 5e-6, "dW of a GraphConv used twice")
    close(plain[1], db1[0] + db2[0], 0.0, 5e-6, "dbias of a GraphConv used twice")


def test_deferral_sees_a_parameter_that_also_feeds_a_plain_torch_op():
    """a kernel that is also read by a torch op in the same graph (an L2 penalty here) receives a second
    contribution that autograd adds INSIDE the backward pass -- its weight gradient must not wait for the flush.  The same for a
    parameter with a tensor hook (the hook reads the gradient inside the pass)."""
    from kgcn_amd import layers
    rng = np.random.default_rng(12)
    m, din, dout = 20000, 256, 256                    # the wide-layer weight-gradient route (deferral-capable)
    x = t32(rng.standard_normal((m, din)).astype(np.float32))
    g = t32(rng.standard_normal((m, dout)).astype(np.float32))
    w = torch.nn.Parameter(t32(rng.standard_normal((din, dout)).astype(np.float32) * 0.05))
    b = torch.nn.Parameter(torch.zeros(dout, device=dev()))
    from kgcn_amd import ops
    _grads_with_and_without_deferral([w, b], lambda: (ops.dense(x, w, b, activation="relu") * g).sum() + 0.5 * (w * w).sum())
    seen = []
    h = w.register_hook(lambda gr: seen.append(float(gr.abs().sum())))
    try:
        _grads_with_and_without_deferral([w, b], lambda: (ops.dense(x, w, b, activation="relu") * g).sum())
    finally:
        h.remove()
    assert len(set(seen)) == 1 and seen[0] > 0, seen   # the hook saw the FINISHED gradient every time, deferral on or off


def test_model_py_layer_calls_through_the_kgcn_import_path():
    """The layer-call sequence of example_model/model.py:41-55, written against `import kgcn.layers` exactly as the
    reference writes it (keyword adj=, enabled_node_nums / max_node_num on the BN layer), must run on the HIP path and
    reproduce the model oracle's logits."""
    import kgcn.layers
    from kgcn_amd import models
    z = load_golden("g3_synthetic_feed_b30.npz")
    adjs = unflatten_adjs(z, "adj_")
    x = z["features"]
    p = M.init_params(np.random.default_rng(6), 3)
    net = {"conv": [kgcn.layers.GraphConv(50, 1) for _ in range(3)], "bn": kgcn.layers.GraphBatchNormalization(),
           "dense": kgcn.layers.GraphDense(50), "out": models.KerasDense(2)}

    def build_model(features, adj):
        layer = features
        layer = net["conv"][0](layer, adj=adj)
        layer = torch.sigmoid(layer)
        layer = net["conv"][1](layer, adj=adj)
        layer = torch.sigmoid(layer)
        layer = net["conv"][2](layer, adj=adj)
        layer = net["bn"](layer, max_node_num=features.shape[1], enabled_node_nums=None)
        layer = torch.sigmoid(layer)
        layer = net["dense"](layer)
        layer = torch.sigmoid(layer)
        layer = kgcn.layers.GraphGather()(layer)
        return net["out"](layer)

    build_model(t32(x), adjs)                          # Keras build semantics: parameters appear on the first call
    with torch.no_grad():
        for i, conv in enumerate(net["conv"], 1):
            conv.w[0].copy_(t32(p["w%d" % i][0])); conv.bias[0].copy_(t32(p["b%d" % i][0]))
        net["bn"].gamma.copy_(t32(p["gamma"])); net["bn"].beta.copy_(t32(p["beta"]))
        net["dense"].kernel.copy_(t32(p["dk"])); net["dense"].bias.copy_(t32(p["db"]))
        net["out"].kernel.copy_(t32(p["ok"])); net["out"].bias.copy_(t32(p["ob"]))
    logits = build_model(t32(x), adjs)
    c = M.forward(p, x.astype(np.float64), adjs, z["labels"].astype(np.float64), z["mask"].astype(np.float64))
    close(logits, c["logits"], atol=1e-5, what="logits through kgcn.layers")      # measured 4.0e-7
    assert type(net["conv"][0]).__module__ == "kgcn_amd.layers"


_DP_GRAPH_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from kgcn_amd import data_util as D, models, train, parallel
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
raw = np.load(os.path.join(sys.argv[1], "tests", "golden", "g1_synthetic_raw.npz"))
chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
ds = D.DeviceGraphDataset(chans, raw["feature"], device=dev)
torch.manual_seed(0)
m_e, m_g = models.GCN(1).to(dev), models.GCN(1).to(dev)
idx = np.arange(30)
adj0, x0 = ds.batch(idx, 30)
m_e(x0, adj0); m_g(x0, adj0)
m_g.load_state_dict(m_e.state_dict())
lab = torch.from_numpy(raw["label"][idx].astype(np.float32)).to(dev)
mask = torch.ones(30, device=dev)
o_e = train.TFAdam(m_e.parameters(), lr=0.01)
o_g = train.TFAdam(m_g.parameters(), lr=0.01, capturable=True)
bucket = parallel.GradBucket(list(m_g.parameters()))
sb = ds.static_batch(30); sb.load(idx)
step = train.GraphedTrainStep(m_g, o_g, models.masked_softmax_ce, sb, lab, mask, bucket=bucket,
                              shard_weight=parallel.shard_weight(30, 30))
out = []
for it in range(4):
    o_e.zero_grad()
    cost_opt, cost_sum = models.masked_softmax_ce(m_e(x0, adj0), lab, mask)
    cost_opt.backward()
    o_e.step()
    cs_g, _ = step.replay()
    out.append([float(cost_sum), float(cs_g)])
dmax = max(float((a - b).abs().max()) for a, b in zip(m_e.parameters(), m_g.parameters()))
torch.cuda.synchronize()
print("RESULT " + json.dumps({"costs": out, "param_diff": dmax}))
dist.destroy_process_group()
"""


def test_graphed_step_with_captured_rccl_allreduce(tmp_path):
    """Data-parallel step as ONE hipGraph: the bucket all-reduce (RCCL, a 1-rank group on this 1-GPU box) sits inside
    the capture between backward and the Adam update (flat bucket, shard weight 1 for the single rank) -- replaying it
    follows eager single-process steps."""
    import json, socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "dp_graph.py"
    script.write_text(_DP_GRAPH_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for attempt in range(3):                     # a rendezvous on a just-freed port can fail transiently: not what is tested here
        r = subprocess.run([sys.executable, str(script), ROOT, str(port)], capture_output=True, text=True, timeout=600, env=env)
        if r.returncode == 0:
            break
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    for ce, cg in res["costs"]:
        assert abs(ce - cg) < 2e-4 * max(1.0, abs(ce)), res
    assert res["costs"][-1][1] < res["costs"][0][1]
    assert res["param_diff"] < 2e-5, res


@pytest.mark.parametrize("route", [1, 2])
@pytest.mark.parametrize("B,N,F,W,with_en", [(30, 10, 3, 50, False), (64, 32, 40, 48, True), (1, 5, 7, 13, True), (200, 16, 5, 50, True),
                                             (37, 17, 64, 50, True), (1500, 10, 3, 50, True), (301, 7, 9, 64, True),
                                             (2000, 10, 3, 50, True)])
def test_cross_layer_stack_equals_layer_by_layer(B, N, F, W, with_en, route):
    """example_model/model.py's node-level body (3 x GraphConv, BatchNormalization with its moving statistics, GraphDense,
    GraphGather) through the cross-layer kernels (one forward, one backward launch; route 1: csrc/stack.hip, one graph per
    workgroup trip, plain FMAs; route 2: csrc/stack_tile.hip, 64-row tiles of whole graphs on the f32 MFMA -- a ragged last tile in
    the 1,500-graph case (250 tiles), more tiles than workgroups (334: second trips) in the 2,000-graph case) against the same modules run one by one: logits, d features and every parameter gradient; non-trivial moving statistics / gamma / beta /
    biases, true sizes (padded rows -> act(0) behind the normalisation), a dummy graph, 32 nodes / odd node counts / 64 input features.  The layer-by-layer route is itself checked against the fp64 model oracle above."""
    from kgcn_amd import layers, models
    from test_oracle_model import tox21_like_batch
    rng = np.random.default_rng(B + N)
    x, adjs, _, mask, _, sizes = tox21_like_batch(rng, B=B, N=N, F=F, T=2)
    if B == 1:
        sizes = np.array([N - 1]); mask = np.ones(1)
        a = K.synth_mol_graphs(rng, 1, N - 1, 1)[0][0]
        adjs = [[(a[0], a[1], [N, N])]]
        x = np.zeros((1, N, F)); x[0, :N - 1] = rng.standard_normal((N - 1, F))
    lab = np.eye(2)[rng.integers(0, 2, B)]
    res = {}
    from kgcn_amd import ops
    for fused in (True, False):
        layers.stack_fusion = fused
        ops.stack_route = route
        max_rows, layers.stack_fusion_max_rows = layers.stack_fusion_max_rows, 1 << 30
        try:
            torch.manual_seed(0)
            model = models.GCN(1, 2).to(dev())
            for m, w in ((model.conv1, W), (model.conv2, W), (model.conv3, W), (model.dense, W)):
                m.output_dim = w
            tx = t32(x).requires_grad_(True)
            en = torch.as_tensor(sizes) if with_en else None
            model(tx, adjs, enabled_node_nums=en)                      # Keras-style build (layer by layer)
            with torch.no_grad():
                gen = torch.Generator(device="cpu").manual_seed(1)
                for p_ in model.parameters():
                    if p_.dim() == 1 or p_.shape[0] == 1:
                        p_.copy_(torch.randn(p_.shape, generator=gen).to(p_.device) * 0.2 + (1.0 if p_ is model.bn.gamma else 0.0))
                model.bn.moving_mean.copy_(torch.randn(W, generator=gen).to(dev()) * 0.1)
                model.bn.moving_variance.copy_(torch.rand(W, generator=gen).to(dev()) + 0.5)
            logits = model(tx, adjs, enabled_node_nums=en)
            used = logits.grad_fn.next_functions[0][0].__class__.__name__ if fused else ""
            cost, _ = models.masked_softmax_ce(logits, t32(lab), t32(mask))
            cost.backward()
            res[fused] = (logits.detach().cpu().numpy(), tx.grad.cpu().numpy(),
                          [(n_, p_.grad.cpu().numpy()) for n_, p_ in model.named_parameters()])
            if fused:                                                  # the fused route really ran
                names = set()
                fn, stack_ = logits.grad_fn, [logits.grad_fn]
                while stack_:
                    f = stack_.pop()
                    if f is None:
                        continue
                    names.add(f.__class__.__name__)
                    stack_ += [nf[0] for nf in f.next_functions]
                assert any("GcnStack" in n_ for n_ in names), names
        finally:
            layers.stack_fusion = True
            layers.stack_fusion_max_rows = max_rows
            ops.stack_route = 0
    close(res[True][0], res[False][0], atol=2e-5, what="stack vs layers: logits")
    close(res[True][1], res[False][1], atol=1e-7, rel=2e-5, what="stack vs layers: d features")
    for (n_, a), (_, b) in zip(res[True][2], res[False][2]):
        close(a, b, atol=2e-7, rel=2e-5, what="stack vs layers: grad %s" % n_)


@pytest.mark.parametrize("T,nsel", [(30, 30), (300, 257), (1000, 1000), (5, 0), (6000, 4096)])
def test_batch_assemble_equals_per_container_gather(T, nsel):
    """kgcn_batch_assemble (all containers of a mini-batch + feature rows + registered tables in two launches) against the
    per-container kgcn_csr_gather_graphs and torch.index_select: bit-exact rowptr / entries / slot tables / graph_ptr of A,
    A^T and both row-padded copies, feature rows, a float and an int32 table; dummy graphs (-1) pad a short batch; more than
    256 graphs (several scan blocks) and an empty selection."""
    from kgcn_amd import data_util as D
    raw = load_golden("g1_synthetic_raw.npz")
    rep = 6
    dense = np.tile(raw["dense_adj"].astype(np.int64), (rep, 1, 1))
    feats = np.tile(raw["feature"], (rep, 1, 1)).astype(np.float32)
    G = dense.shape[0]
    chans, _ = D.build_adjs({"dense_adj": dense, "max_node_num": 10})
    ds = D.DeviceGraphDataset(chans, feats, device=dev())
    rng = np.random.default_rng(T)
    lab = torch.from_numpy(rng.standard_normal((G, 3)).astype(np.float32)).to(dev())
    sizes = torch.from_numpy(rng.integers(0, 11, G).astype(np.int32)).to(dev())
    sb = ds.static_batch(T)
    lab_s, sz_s = sb.add_table(lab), sb.add_table(sizes)
    idx = rng.integers(0, G, nsel)
    sb.load(idx)
    sel = np.full(T, -1, np.int64); sel[:nsel] = idx
    for pairs in sb._sources:
        for src, st in pairs:
            ref = src.gather(sel)
            n = int(ref.rowptr[-1])
            assert torch.equal(st.rowptr, ref.rowptr), "rowptr (row_pad %d)" % src.row_pad
            assert torch.equal(st.cv[:n], ref.cv[:n]), "entries (row_pad %d)" % src.row_pad
            assert int(st._gptr_buf[-1]) == n
            if src.row_pad:
                assert torch.equal(st.slots, ref.slots) and torch.equal(st._gptr_buf, ref.graph_ptr)
    seld = torch.from_numpy(np.maximum(sel, 0)).to(dev())
    valid = torch.from_numpy(sel >= 0).to(dev())
    assert torch.equal(sb.features, ds.features[seld] * valid[:, None, None])
    assert torch.equal(lab_s, lab[seld] * valid[:, None])
    assert torch.equal(sz_s, sizes[seld] * valid.to(torch.int32))


def test_graphed_train_step_with_captured_assembly():
    """GraphedTrainStep(capture_assembly=True): the device-side batch assembly (adjacency, features, label / mask tables) is the
    head of the hipGraph and a step is stage(indices) + replay(); costs and parameters follow eager steps on batches assembled
    the plain way, through two epochs with a padded last batch."""
    from kgcn_amd import data_util as D, models, train
    raw = load_golden("g1_synthetic_raw.npz")
    chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
    feats, labels = raw["feature"], raw["label"]
    G = feats.shape[0]
    ds = D.DeviceGraphDataset(chans, feats, device=dev())
    torch.manual_seed(0)
    m_e, m_g = models.GCN(1).to(dev()), models.GCN(1).to(dev())
    adj0, x0 = ds.batch(np.arange(30), 30)
    m_e(x0, adj0); m_g(x0, adj0)
    m_g.load_state_dict(m_e.state_dict())
    o_e = train.TFAdam(m_e.parameters(), lr=0.01)
    o_g = train.TFAdam(m_g.parameters(), lr=0.01)
    sb = ds.static_batch(30)
    lab_s = sb.add_table(t32(labels))
    mask_s = sb.add_table(torch.ones(G, device=dev()))
    sb.load(np.arange(30))
    step = train.GraphedTrainStep(m_g, o_g, models.masked_softmax_ce, sb, lab_s, mask_s, capture_assembly=True)
    rng = np.random.default_rng(2)
    idx = np.arange(160)
    lab, mask = torch.zeros((30, 2), device=dev()), torch.zeros(30, device=dev())
    for epoch in range(2):
        rng.shuffle(idx)
        for it in range(6):
            bidx = idx[it * 30:(it + 1) * 30]
            nb = len(bidx)
            lab.zero_(); lab[:nb] = t32(labels[bidx])
            mask.zero_(); mask[:nb] = 1
            adj, x = ds.batch(bidx, 30)
            cs_e, lg_e = train.train_step(m_e, o_e, models.masked_softmax_ce, x, adj, lab, mask)
            sb.stage(bidx)
            cs_g, lg_g = step.replay()
            assert abs(cs_e - float(cs_g)) < 1e-4 * max(1.0, abs(cs_e)), (epoch, it, cs_e, float(cs_g))
            close(lg_g, lg_e.cpu().numpy(), atol=1e-6, what="logits")                # measured 0.0: the replay launches the same kernels
    for a, b in zip(m_e.parameters(), m_g.parameters()):
        close(b, a.detach().cpu().numpy(), atol=1e-7, what="params after 12 steps")


def test_weight_tables_refreshed_once_per_step_follow_the_weights():
    """ops.weight_tables: the W / W^T fragment tables of wide dense layers split by ONE launch at the start of a step must
    give the same results as the per-call split, and must never be used stale: after an optimiser update (raw-pointer write,
    epoch bump), after a torch-level in-place change of the weight (version bump), and inside a captured hipGraph whose weights
    were restored behind its back, outputs and gradients follow the CURRENT weights."""
    from kgcn_amd import ops, train
    torch.manual_seed(3)
    m, din, dout = 4096, 256, 256
    x = torch.randn(m, din, device=dev(), requires_grad=True)
    lin1 = torch.nn.Parameter(torch.randn(din, dout, device=dev()) * 0.05)
    lin2 = torch.nn.Parameter(torch.randn(dout, dout, device=dev()) * 0.05)

    def run():
        for t in (x, lin1, lin2):
            t.grad = None
        y = ops.dense(ops.dense(x, lin1, None, "relu"), lin2, None, "sigmoid")
        y.square().sum().backward()
        return y.detach().clone(), x.grad.clone(), lin1.grad.clone(), lin2.grad.clone()

    ops.enabled_weight_tables = False
    try:
        ref = run()
    finally:
        ops.enabled_weight_tables = True
    run()                                            # registers the two weights
    ops.weight_tables.refresh()
    assert ops.weight_tables.lookup(lin1, 0)[0] is not None and ops.weight_tables.lookup(lin2, 1)[0] is not None
    got = run()
    for a, b in zip(got, ref):
        assert torch.equal(a, b)                     # same kernels, same tables: bit-identical
    # torch-level in-place change: the version counter invalidates the tables until the next refresh
    with torch.no_grad():
        lin1.mul_(1.5)
    assert ops.weight_tables.lookup(lin1, 0)[0] is None and ops.weight_tables.lookup(lin2, 0)[0] is not None
    ops.enabled_weight_tables = False
    try:
        ref2 = run()
    finally:
        ops.enabled_weight_tables = True
    got2 = run()
    for a, b in zip(got2, ref2):
        assert torch.equal(a, b)
    # an optimiser update writes through raw pointers: its epoch bump invalidates everything
    ops.weight_tables.refresh()
    opt = train.TFAdam([lin1, lin2], lr=0.1)
    ops.weight_tables.refresh()
    run()
    opt.step()
    assert ops.weight_tables.lookup(lin1, 0)[0] is None and ops.weight_tables.lookup(lin2, 1)[0] is None
    ops.enabled_weight_tables = False
    try:
        ref3 = run()
    finally:
        ops.enabled_weight_tables = True
    ops.weight_tables.refresh()
    got3 = run()
    for a, b in zip(got3, ref3):
        assert torch.equal(a, b)


@pytest.mark.parametrize("route", [1, 2])
def test_cross_layer_stack_rows_with_duplicate_entries(route):
    """Rows that store a column several times (more entries than nodes: the tile kernels leave their ELL copy for a lockstep
    walk over the CSR) accumulate like the layer-by-layer route; small adjacency values keep the sigmoids out of saturation so
    the comparison stays well conditioned."""
    from kgcn_amd import layers, models, ops
    rng = np.random.default_rng(11)
    B, N, F, W = 70, 6, 9, 24
    adjs, x = [], rng.standard_normal((B, N, F)).astype(np.float32)
    for b in range(B):
        a = (rng.random((N, N)) < 0.6) * rng.standard_normal((N, N)) * 0.2
        ix, vl = np.argwhere(a != 0).astype(np.int32), a[a != 0].astype(np.float32)
        k = rng.integers(0, len(ix), size=3 * len(ix))
        ix, vl = np.concatenate([ix, ix[k]]), np.concatenate([vl, vl[k]])
        p = rng.permutation(len(ix))
        adjs.append([(ix[p], vl[p], [N, N])])
    assert max(np.bincount(a[0][0][:, 0], minlength=N).max() for a in adjs) > N
    res = {}
    for fused in (True, False):
        layers.stack_fusion, ops.stack_route = fused, route
        try:
            torch.manual_seed(0)
            model = models.GCN(1, 2).to(dev())
            for m in (model.conv1, model.conv2, model.conv3, model.dense):
                m.output_dim = W
            tx = t32(x).requires_grad_(True)
            model(tx, adjs)
            logits = model(tx, adjs)
            logits.square().sum().backward()
            res[fused] = [logits.detach().cpu().numpy(), tx.grad.cpu().numpy()] + [p_.grad.cpu().numpy() for p_ in model.parameters()]
        finally:
            layers.stack_fusion, ops.stack_route = True, 0
    for a, b in zip(res[True], res[False]):
        close(a, b, atol=2e-6, rel=5e-5, what="stack with duplicate entries vs layers")


@pytest.mark.parametrize("phase", [0, 1])
def test_deepchem_model_against_the_oracle_ops(phase):
    """example_model/model_deepchem.py (GraphConv -> relu -> GraphMaxPooling -> GraphBatchNormalization over the valid rows, four
    times; GraphDense -> sigmoid -> GraphGather -> Dense) on Tox21-shaped molecules with true sizes, in both Keras learning
    phases: logits against the same chain of oracle ops in fp64; gradients flow to every parameter."""
    from kgcn_amd import layers, models
    from test_oracle_model import tox21_like_batch
    rng = np.random.default_rng(3 + phase)
    x, adjs, _, mask, _, sizes = tox21_like_batch(rng, B=24, N=20, F=17, T=2)
    layers.set_learning_phase(phase)
    try:
        torch.manual_seed(0)
        model = models.DeepChemGCN(1, 2).to(dev())
        tx = t32(x)
        model(tx, adjs, enabled_node_nums=torch.as_tensor(sizes))                      # Keras-style build
        with torch.no_grad():
            gen = torch.Generator(device="cpu").manual_seed(5)
            for bn in model.bn:
                bn.gamma.copy_(torch.rand(bn.gamma.shape, generator=gen).to(dev()) + 0.5)
                bn.beta.copy_(torch.randn(bn.beta.shape, generator=gen).to(dev()) * 0.2)
                bn.moving_mean.copy_(torch.randn(bn.moving_mean.shape, generator=gen).to(dev()) * 0.1)
                bn.moving_variance.copy_(torch.rand(bn.moving_variance.shape, generator=gen).to(dev()) + 0.5)
            stats = [(bn.moving_mean.cpu().numpy().astype(np.float64), bn.moving_variance.cpu().numpy().astype(np.float64)) for bn in model.bn]
        logits = model(tx, adjs, enabled_node_nums=torch.as_tensor(sizes))
        logits.square().sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    finally:
        layers.set_learning_phase(0)
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    h = x.astype(np.float64)
    for conv, bn, (mm, mv) in zip(model.conv, model.bn, stats):
        h = np.maximum(K.graphconv_fwd(h, adjs, [f64(conv.w[0])], [f64(conv.bias[0])]), 0.0)
        h = K.graph_maxpool_fwd(h, adjs)
        h = K.graph_bn_fwd(h, f64(bn.gamma), f64(bn.beta), mm, mv, enabled_node_nums=sizes, training=bool(phase))[0]
    B, N, D = h.shape
    h = 1.0 / (1.0 + np.exp(-(h.reshape(B * N, D) @ f64(model.dense.kernel) + f64(model.dense.bias)))).reshape(B, N, -1)
    ref = h.sum(1) @ f64(model.out.kernel) + f64(model.out.bias)
    close(logits, ref, atol=1e-6, rel=6e-6, what="model_deepchem logits (phase %d)" % phase)     # measured 6.1e-7 of |ref|max


def test_node_label_model_against_the_oracle_ops():
    """example_model/model_node_label.py: two GraphConv(64) -> GraphBatchNormalization (valid rows) -> relu blocks and a
    GraphConv(2) read per NODE; softmax cross entropy averaged over the node rows of a graph, masked per graph: logits and both
    cost values against the same chain of oracle ops in fp64; the 2-wide last GraphConv goes through the narrow kernels."""
    from kgcn_amd import models
    from test_oracle_model import tox21_like_batch
    rng = np.random.default_rng(9)
    x, adjs, _, mask, _, sizes = tox21_like_batch(rng, B=20, N=15, F=12, T=2)
    node_lab = np.eye(2)[rng.integers(0, 2, (20, 15))]
    torch.manual_seed(0)
    model = models.NodeLabelGCN(1, 2).to(dev())
    tx = t32(x)
    model(tx, adjs, enabled_node_nums=torch.as_tensor(sizes))
    with torch.no_grad():
        gen = torch.Generator(device="cpu").manual_seed(2)
        for bn in model.bn:
            bn.gamma.copy_(torch.rand(bn.gamma.shape, generator=gen).to(dev()) + 0.5)
            bn.beta.copy_(torch.randn(bn.beta.shape, generator=gen).to(dev()) * 0.2)
            bn.moving_mean.copy_(torch.randn(bn.moving_mean.shape, generator=gen).to(dev()) * 0.1)
            bn.moving_variance.copy_(torch.rand(bn.moving_variance.shape, generator=gen).to(dev()) + 0.5)
        for conv in model.conv:
            conv.bias[0].copy_(torch.randn(conv.bias[0].shape, generator=gen).to(dev()) * 0.1)
    logits = model(tx, adjs, enabled_node_nums=torch.as_tensor(sizes))
    cost_opt, cost_sum = models.node_softmax_ce(logits, t32(node_lab), t32(mask))
    cost_opt.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    h = x.astype(np.float64)
    for conv, bn in zip(model.conv[:2], model.bn):
        h = K.graphconv_fwd(h, adjs, [f64(conv.w[0])], [f64(conv.bias[0])])
        h = np.maximum(K.graph_bn_fwd(h, f64(bn.gamma), f64(bn.beta), f64(bn.moving_mean), f64(bn.moving_variance),
                                      enabled_node_nums=sizes)[0], 0.0)
    ref = K.graphconv_fwd(h, adjs, [f64(model.conv[2].w[0])], [f64(model.conv[2].bias[0])])
    close(logits, ref, atol=1e-6, rel=2e-6, what="model_node_label logits")               # measured 1.4e-7
    lse = np.log(np.exp(ref - ref.max(2, keepdims=True)).sum(2)) + ref.max(2)
    ce = -(node_lab * (ref - lse[..., None])).sum(2)
    cost = mask * ce.mean(1)
    assert abs(float(cost_opt) - cost.mean()) < 2e-5 and abs(float(cost_sum) - cost.sum()) < 2e-4
