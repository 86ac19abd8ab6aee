#!/usr/bin/env python3
"""kgcn train --config example_config/synth.json, on the MI355X path: example_model/model.py's network on the
reference's synthetic dataset (the copy kept as a test fixture: tests/golden/g1_synthetic_raw.npz = the arrays of
example_jbl/synthetic.jbl), batch 30, learning rate 1e-3, 160 / 40 train / validation split, TF-style Adam.
The dataset lives in HBM; every mini-batch is assembled on the device and the whole step (forward, loss, backward,
optimiser) is one hipGraph replay.

    python examples/train_synthetic.py [epochs]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, models, train  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
raw = np.load(os.path.join(ROOT, "tests", "golden", "g1_synthetic_raw.npz"))
channels, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": int(raw["max_node_num"])})
dataset = D.DeviceGraphDataset(channels, raw["feature"], device=dev)
labels_all = torch.from_numpy(raw["label"].astype(np.float32)).to(dev)
BATCH = 30
train_idx, valid_idx = np.arange(160), np.arange(160, 200)

torch.manual_seed(0)
model = models.GCN(adj_channel_num=len(channels)).to(dev)
adj0, x0 = dataset.batch(train_idx[:BATCH], BATCH)
model(x0, adj0)                                                   # creates the parameters (Keras-style lazy build)
opt = train.TFAdam(model.parameters(), lr=1e-3, capturable=True)
batch = dataset.static_batch(BATCH)
labels = batch.add_table(labels_all)                              # the selected graphs' label rows ...
mask = batch.add_table(torch.ones(len(labels_all), device=dev))   # ... and 1 per real graph, 0 per dummy: same assembly launch
batch.load(train_idx[:BATCH])
# the device-side assembly (adjacency, features, labels, mask) is the head of the captured step: one step = stage() + replay()
step = train.GraphedTrainStep(model, opt, models.masked_softmax_ce, batch, labels, mask, capture_assembly=True)


def fill(idx):
    batch.stage(idx)                                              # one small pinned upload of the selected graph indices
    return len(idx)


rng = np.random.default_rng(1234)
for epoch in range(epochs):
    rng.shuffle(train_idx)
    cost = 0.0
    for it in range(0, len(train_idx), BATCH):
        fill(train_idx[it:it + BATCH])                            # the last batch holds 10 real + 20 dummy graphs
        cs, _ = step.replay()
        cost += float(cs)
    with torch.no_grad():
        correct = 0
        for it in range(0, len(valid_idx), BATCH):
            idx = valid_idx[it:it + BATCH]
            adj, x = dataset.batch(idx, BATCH)
            pred = model(x, adj)[:len(idx)].argmax(1)
            correct += int((pred == labels_all[torch.as_tensor(idx, device=dev)].argmax(1)).sum())
    print("epoch %2d  training cost %.4f  validation accuracy %.3f" % (epoch, cost / len(train_idx), correct / len(valid_idx)))
