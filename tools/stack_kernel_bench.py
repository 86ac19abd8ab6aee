#!/usr/bin/env python3
"""HIP-event times of the cross-layer stack launches alone (forward, backward) per route and batch size, model.py body on
10-node synthetic graphs.  usage: stack_kernel_bench.py [batch ...]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, layers, models, ops  # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "g1_synthetic_raw.npz"))
REP = 100
dense = np.tile(z["dense_adj"].astype(np.int64), (REP, 1, 1))
feats = np.tile(z["feature"], (REP, 1, 1)).astype(np.float32)
chans, _ = D.build_adjs({"dense_adj": dense, "max_node_num": 10})
ds = D.DeviceGraphDataset(chans, feats, device=dev)
layers.stack_fusion_max_rows = 1 << 30
res = {}


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


for batch in [int(a) for a in sys.argv[1:]] or [30, 512, 1024, 4096]:
    row = {}
    for route in (1, 2):
        ops.stack_route = route
        torch.manual_seed(0)
        model = models.GCN(1).to(dev)
        sb = ds.static_batch(batch)
        sb.load(np.arange(batch))
        model(sb.features, sb.adjacency)
        state = {}

        def fwd():
            state["y"] = model(sb.features, sb.adjacency)

        def fwd_bwd():
            y = model(sb.features, sb.adjacency)
            y.sum().backward()

        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fwd_bwd()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fwd()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            fwd_bwd()
        tf = timed(g.replay)
        tfb = timed(g2.replay)
        row["route%d" % route] = {"model_fwd_us": tf, "model_fwd_bwd_us": tfb}
    res["batch_%d" % batch] = row
print(json.dumps(res, indent=1))
