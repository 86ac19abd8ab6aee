#!/usr/bin/env python3
"""dev: which weight-gradient calls wait for the flush in one eager GCN-split step, and the error each choice leaves"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kgcn_amd import data_util as D, models, train, ops
raw = np.load(os.path.join(ROOT, "tests/golden/g1_synthetic_raw.npz"))
dev = torch.device("cuda:0")
chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10}, split_adj_flag=True)
ds = D.DeviceGraphDataset(chans, raw["feature"], device=dev)
C = len(chans)
torch.manual_seed(1)
m_a, m_b = models.GCN(C).to(dev), models.GCN(C).to(dev)
idx = np.arange(30)
adj0, x0 = ds.batch(idx, 30)
m_a(x0, adj0); m_b(x0, adj0)
m_b.load_state_dict(m_a.state_dict())
lab = torch.tensor(raw["label"][idx].astype(np.float32), device=dev)
mask = torch.ones(30, device=dev)
orig = ops._single_use
def traced(*ps):
    r = orig(*ps)
    print("  single_use", [None if p is None else (tuple(p.shape), ops._param_uses.get(id(p), 0), p.is_leaf) for p in ps], "->", r)
    return r
ops._single_use = traced
def grads(m, defer):
    os.environ["KGCN_NO_DEFER"] = "" if defer else "1"
    for p in m.parameters():
        p.grad = None
    ops.weight_tables.refresh()
    lg = m(x0, adj0)
    c, _ = models.masked_softmax_ce(lg, lab, mask)
    with ops.deferred_reductions():
        c.backward()
    torch.cuda.synchronize()
    return [(n, p.grad.clone()) for n, p in m.named_parameters()]
for trial in range(3):
    print("trial", trial)
    ga = grads(m_a, True)
    gb = grads(m_b, False)
    for (n, a), (_, b) in zip(ga, gb):
        print("  %-40s %s  max|d| %.3e  of %.3e" % (n, tuple(a.shape), float((a - b).abs().max()), float(b.abs().max())))
