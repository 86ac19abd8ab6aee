"""Model-level parity: example_model/model.py's network on synthetic.jbl batches,
HIP path (fp32) vs the fp64 model oracle -- logits, loss, every parameter gradient, and the
parameter trajectory over several TF-Adam steps; plus a short training run (loss must fall) through
the product's own loaders for GCN and GIN."""
import numpy as np
import pytest
import torch

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_model_oracle as M
from test_gpu_parity import close, dev, t32

pytestmark = pytest.mark.gpu


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
            close(logits, c["logits"], atol=2e-5, what="logits")
            assert abs(float(cost_opt) - c["cost_opt"]) < 5e-6 and abs(float(cost_sum) - c["cost_sum"]) < 1e-4
            for k, t in _named(model).items():
                ref = g[k][0] if isinstance(g[k], list) else g[k]
                close(t.grad, np.asarray(ref).reshape(tuple(t.shape)), atol=1e-6, rel=1e-5, what="grad " + k)
        opt.step()
        p = oopt.step(p, g)
    for k, t in _named(model).items():               # parameters after 6 Adam steps
        ref = p[k][0] if isinstance(p[k], list) else p[k]
        close(t, np.asarray(ref).reshape(tuple(t.shape)), atol=2e-4, what="param " + k)


@pytest.mark.parametrize("kind", ["GCN", "GIN"])
def test_short_training_run_on_synthetic_jbl(kind):
    """kgcn train --config example_config/synth.json in miniature: batch 30, padded last batch,
    shuffled epochs, TF Adam; through the product's loaders.  The loss must fall."""
    from kgcn_amd import data_util as D, models, train
    raw = load_golden("g1_synthetic_raw.npz")
    chans, enabled = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
    feats, labels = raw["feature"], raw["label"]
    model = (models.GCN(1) if kind == "GCN" else models.GIN(1)).to(dev())
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
