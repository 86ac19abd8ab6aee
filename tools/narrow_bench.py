#!/usr/bin/env python3
"""Narrow layers of cfg4's tail (T=4096, N=50): the fused GraphConv kernels against dense + aggregation at the same
widths, forward and forward+backward, with the algorithmic HBM time next to each."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from config_bench_util import mol_batch  # noqa
from kgcn_amd import ops
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
for din, dout in ((50, 50), (64, 64), (32, 50)):
    x = torch.randn(B, N, din, device=dev, requires_grad=True)
    w = torch.randn(din, dout, device=dev, requires_grad=True)
    b = torch.randn(dout, device=dev, requires_grad=True)
    g = torch.randn(B, N, dout, device=dev)
    fl = B * N * 4 * (din + dout) / 8e12 * 1e6
    key = "%d->%d" % (din, dout)
    if ops.graphconv_fused_supported(csr, din, dout):
        def ff(): return ops.graphconv_fused(x, w, b, csr)
        def fb():
            x.grad = w.grad = b.grad = None
            ff().backward(g)
        res[key + " fused fwd"] = {"us": round(timed(ff), 1), "hbm_floor_us": round(fl, 1)}
        res[key + " fused fwd+bwd"] = {"us": round(timed(fb), 1), "hbm_floor_us": round(fl + B * N * 4 * (2 * din + dout) / 8e12 * 1e6, 1)}
    def df(): return ops.dense(x.reshape(-1, din), w, b)
    def db():
        x.grad = w.grad = b.grad = None
        df().backward(g.reshape(-1, dout))
    res[key + " dense fwd"] = {"us": round(timed(df), 1), "hbm_floor_us": round(fl, 1)}
    res[key + " dense fwd+bwd"] = {"us": round(timed(db), 1)}
print(json.dumps(res, indent=1))
