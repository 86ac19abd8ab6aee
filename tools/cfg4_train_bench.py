#!/usr/bin/env python3
"""BASELINE config 4 end to end on ONE GPU: Tox21-shaped multitask training (example_model/model_multitask.py,
example_config/multitask.json shape: 12 tasks, N = 50 padded with variable true sizes, F = 81, masked labels),
dataset resident in HBM, batch 4096 assembled on the device, forward + masked sigmoid CE + backward + TF-Adam.
Synthetic molecules (random tree + extra edges + self loops on the first `size` nodes, Kipf-normalised values).
usage: python tools/cfg4_train_bench.py [graphs=200000] [batch=4096] [steps=30]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, models, train  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 30
N, F, T = 50, 81, 12
dev = torch.device("cuda:0")
rng = np.random.default_rng(4)

t_gen = time.perf_counter()
sizes = rng.integers(5, N + 1, size=G)
A = np.zeros((G, N, N), np.bool_)
ar = np.arange(G)
for i in range(1, N):                                   # random tree over the first `size` nodes of every graph
    act = sizes > i
    j = (rng.random(G) * i).astype(np.int64)
    A[ar[act], i, j[act]] = True
    A[ar[act], j[act], i] = True
for _ in range(2):                                      # two extra edges per molecule
    i = (rng.random(G) * sizes).astype(np.int64)
    j = (rng.random(G) * sizes).astype(np.int64)
    A[ar, i, j] = True
    A[ar, j, i] = True
node = np.arange(N)
valid = node[None, :] < sizes[:, None]
A[:, node, node] = valid                                # self loops on real nodes only
g, r, c = np.nonzero(A)
del A
chan = D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), G, N)
chan = D.normalize_adj(chan)
feats = (rng.standard_normal((G, N, F)).astype(np.float32)) * valid[:, :, None]
labels = (rng.random((G, T)) < 0.3).astype(np.float32)
mask_label = (rng.random((G, T)) < 0.8).astype(np.float32)
gen_s = time.perf_counter() - t_gen

ds = D.DeviceGraphDataset([chan], feats, device=dev)
lab_d, ml_d = torch.from_numpy(labels).to(dev), torch.from_numpy(mask_label).to(dev)
sizes_d = torch.from_numpy(sizes).to(dev)
torch.manual_seed(0)
model = models.MultitaskGCN(1, T).to(dev)
idx0 = np.arange(B)
adj0, x0 = ds.batch(idx0, B)
model(x0, adj0, enabled_node_nums=sizes_d[:B])
opt = train.TFAdam(model.parameters(), lr=1e-3)
mask = torch.ones(B, device=dev)


def run(steps):
    perm = rng.permutation(G)
    tot = 0.0
    for s in range(steps):
        idx = perm[(s * B) % (G - B):(s * B) % (G - B) + B]
        it = torch.from_numpy(idx).to(dev)
        adj, x = ds.batch(idx, B)
        opt.zero_grad()
        logits = model(x, adj, enabled_node_nums=sizes_d[it])
        cost_opt, cost_sum = models.masked_sigmoid_ce(logits, lab_d[it], mask, ml_d[it])
        cost_opt.backward()
        opt.step()
        tot = cost_sum
    return tot


run(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
last = run(STEPS)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": "cfg4: model_multitask.py, %d graphs resident, batch %d, N=%d (true sizes 5..50), F=%d, %d tasks" % (G, B, N, F, T),
                  "ms_per_step": dt / STEPS * 1e3, "graphs_per_s": B * STEPS / dt, "steps": STEPS,
                  "dataset_bytes_hbm": int(feats.nbytes + 8 * g.shape[0] + 4 * (G * N + 1)),
                  "host_generation_s": gen_s, "final_cost_sum": float(last)}, indent=1))
