#!/usr/bin/env python3
"""End-to-end training throughput of example_model/model.py's network (kgcn train --config
example_config/synth.json: GraphConv(50) x3, BN, GraphDense(50), gather, Dense(2); TF-Adam) on
synthetic.jbl-shaped data (10-node graphs, 3 features) replicated to `graphs` graphs: eager steps vs the
hipGraph-captured step, both with device-side batch assembly (GPU box).  Prints one JSON object.
usage: python tools/train_bench.py [batch_size ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, models, train  # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "g1_synthetic_raw.npz"))
REP = 100                                                     # 200 -> 20,000 graphs
dense = np.tile(z["dense_adj"].astype(np.int64), (REP, 1, 1))
feats = np.tile(z["feature"], (REP, 1, 1)).astype(np.float32)
labels = np.tile(z["label"], (REP, 1)).astype(np.float32)
chans, _ = D.build_adjs({"dense_adj": dense, "max_node_num": 10})
ds = D.DeviceGraphDataset(chans, feats, device=dev)
G = dense.shape[0]
lab_all = torch.from_numpy(labels).to(dev)
res = {"graphs": G, "model": "example_model/model.py (GCN)", "n_nodes": 10, "features": 3}


def run(batch, steps):
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    model = models.GCN(1).to(dev)
    adj0, x0 = ds.batch(np.arange(batch), batch)
    model(x0, adj0)
    out = {}
    lab = torch.zeros((batch, 2), device=dev)
    mask = torch.ones(batch, device=dev)
    batches = [rng.integers(0, G, size=batch) for _ in range(steps)]
    idx_dev = [torch.from_numpy(b).to(dev) for b in batches]
    # eager
    opt = train.TFAdam(model.parameters(), lr=1e-3)
    for w in range(3):
        adj, x = ds.batch(batches[w], batch)
        train.train_step(model, opt, models.masked_softmax_ce, x, adj, lab, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b, bi in zip(batches, idx_dev):
        adj, x = ds.batch(b, batch)
        lab.copy_(lab_all.index_select(0, bi))
        opt.zero_grad()
        logits = model(x, adj)
        cost_opt, cost_sum = models.masked_softmax_ce(logits, lab, mask)
        cost_opt.backward()
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["eager"] = {"ms_per_step": 1e3 * dt / steps, "graphs_per_s": batch * steps / dt}
    # no reference to the eager autograd graph may survive into the capture (its AccumulateGrad nodes are
    # bound to the default stream)
    del logits, cost_opt, cost_sum, adj, x
    # hipGraph
    opt_g = train.TFAdam(model.parameters(), lr=1e-3, capturable=True)
    sb = ds.static_batch(batch)
    sb.load(batches[0])
    step = train.GraphedTrainStep(model, opt_g, models.masked_softmax_ce, sb, lab, mask)
    for w in range(3):
        sb.load(batches[w]); step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b, bi in zip(batches, idx_dev):
        sb.load(b)
        lab.copy_(lab_all.index_select(0, bi))
        step.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["hipgraph"] = {"ms_per_step": 1e3 * dt / steps, "graphs_per_s": batch * steps / dt}
    # replay alone (batch resident): the device time of one step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["hipgraph_replay_only"] = {"ms_per_step": 1e3 * dt / steps, "graphs_per_s": batch * steps / dt}
    return out


for bs in [int(a) for a in sys.argv[1:]] or [30, 4096]:
    res["batch_%d" % bs] = run(bs, 200 if bs <= 256 else 60)
print(json.dumps(res, indent=1))
