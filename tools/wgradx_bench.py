#!/usr/bin/env python3
"""development: HIP-event time of kgcn_dense_wgrad_dact_f32 at cfg4's first-layer shape (117,888 x 84 -> 256, sigmoid')."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import _lib
from kgcn_amd._lib import lib, ptr, check, current_stream
m, din, dout = 117888, 84, 256
dev = torch.device("cuda:0")
x = torch.randn(m, din, device=dev); dy = torch.randn(m, dout, device=dev); y = torch.rand(m, dout, device=dev)
dw = torch.empty(din, dout, device=dev); db = torch.empty(dout, device=dev)
wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout) * 4
ws = torch.empty(wsb // 4, device=dev)
def run():
    check(lib.kgcn_dense_wgrad_dact_f32(ptr(x), din, ptr(dy), ptr(y), dout, 1, m, din, dout, ptr(dw), ptr(db), ptr(ws), wsb, current_stream()), "wgrad")
for _ in range(10): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("%s: %.1f us per call (kernel + second stage)" % (os.environ.get("KGCN_WX", "default"), e0.elapsed_time(e1) / 50 * 1e3))
if os.environ.get("KGCN_WX_CHECK"):
    import numpy as np
    xs, gs, ys = x[:20000].double().cpu(), dy[:20000].double().cpu(), y[:20000].double().cpu()
    m2 = 20000
    dw2 = torch.empty(din, dout, device=dev); db2 = torch.empty(dout, device=dev)
    check(lib.kgcn_dense_wgrad_dact_f32(ptr(x), din, ptr(dy), ptr(y), dout, 1, m2, din, dout, ptr(dw2), ptr(db2), ptr(ws), wsb, current_stream()), "wgrad")
    ref = xs.T @ (gs * ys * (1 - ys))
    print("max rel err dW vs fp64 on 20,000 rows: %.2e" % float((dw2.double().cpu() - ref).abs().max() / ref.abs().max()))
