#!/usr/bin/env python3
"""model.py training step (hipGraph replay, batch resident) with and without the cross-layer kernels over batch sizes:
the one-launch-per-direction stack on its two routes (stack: one graph per workgroup trip, plain fp32 FMAs, csrc/stack.hip;
tiles: 64-row tiles on the f32 MFMA, csrc/stack_tile.hip) against the per-layer kernels."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, layers, models, ops, train  # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "g1_synthetic_raw.npz"))
REP = 100
dense = np.tile(z["dense_adj"].astype(np.int64), (REP, 1, 1))
feats = np.tile(z["feature"], (REP, 1, 1)).astype(np.float32)
chans, _ = D.build_adjs({"dense_adj": dense, "max_node_num": 10})
ds = D.DeviceGraphDataset(chans, feats, device=dev)
res = {}
for batch in [int(a) for a in sys.argv[1:]] or [30, 128, 256, 512, 1024, 4096]:
    row = {}
    for name, fused, route in (("stack", True, 1), ("tiles", True, 2), ("layers", False, 0)):
        layers.stack_fusion = fused
        ops.stack_route = route
        layers.stack_fusion_max_rows = 1 << 30
        torch.manual_seed(0)
        model = models.GCN(1).to(dev)
        sb = ds.static_batch(batch)
        sb.load(np.arange(batch))
        model(sb.features, sb.adjacency)
        lab = torch.zeros((batch, 2), device=dev); lab[:, 0] = 1
        mask = torch.ones(batch, device=dev)
        opt = train.TFAdam(model.parameters(), lr=1e-3)
        step = train.GraphedTrainStep(model, opt, models.masked_softmax_ce, sb, lab, mask)
        for _ in range(5):
            step.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            step.replay()
        torch.cuda.synchronize()
        row[name] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
    res["batch_%d" % batch] = row
print(json.dumps(res, indent=1))
