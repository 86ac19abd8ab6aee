#!/usr/bin/env python3
"""Kernel durations (torch profiler) of the 50-wide dense layer calls at m = 204,800: forward, dX, weight gradient."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, check, ptr, current_stream
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
res = {}
for din, dout in ((50, 50), (50, 12), (63, 33)):
    x = torch.randn(M, din, device=dev); w = torch.randn(din, dout, device=dev); b = torch.randn(dout, device=dev)
    y = torch.empty(M, dout, device=dev); dx = torch.empty_like(x)
    dw = torch.empty_like(w); db = torch.empty_like(b)
    wsb = lib.kgcn_dense_wgrad_workspace_bytes(M, din, dout)
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    def run():
        check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, 1, None, 0, current_stream()))
        check(lib.kgcn_dense_fwd_ws_f32(ptr(y), M, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0, None, 0, current_stream()))
        check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(y), dout, M, din, dout, ptr(dw), ptr(db), ptr(ws), wsb, current_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): run()
        torch.cuda.synchronize()
    ref = x.double() @ w.double() + b.double()
    err = float((y.double() - torch.sigmoid(ref)).abs().max())
    res["%dx%d" % (din, dout)] = {"err_fwd": err, "kernels": {e.key[:40]: round(e.device_time_total / e.count, 1) for e in prof.key_averages() if e.device_time_total > 0}}
print(json.dumps(res))
