#!/usr/bin/env python3
"""Aggregation at wide feature widths (cfg4: T=4096, N=50, d=256): the kernel on [T, N, 256] against the same bytes
as [4T, N, 64] (the contiguous shape of the tile kernel) and a device copy of the same size."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from config_bench_util import mol_batch  # noqa
from kgcn_amd import ops, BatchedAdjacency
dev = torch.device("cuda:0")

def timed(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

res = {}
B, N = 4096, 50
_, csr = mol_batch(B, N)
for d in (256, 128, 64, 50):
    x = torch.randn(B, N, d, device=dev)
    us = timed(lambda: ops.bspmm(csr, x))
    byt = 8 * B * N * d
    res["bspmm d=%d" % d] = {"us": round(us, 1), "GBs": round(byt / us / 1e3, 1)}
    y = torch.empty_like(x)
    us = timed(lambda: y.copy_(x))
    res["copy  d=%d" % d] = {"us": round(us, 1), "GBs": round(byt / us / 1e3, 1)}
print(json.dumps(res, indent=1))
