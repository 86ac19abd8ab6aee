#!/usr/bin/env python3
"""One line per shape: forward time of the wide dense layer through the workspace API (route by KGCN_DENSE_ROUTE)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, check, ptr, current_stream
dev = torch.device("cuda:0")
M = 204800
res = {}
for din, dout in ((256, 256), (512, 256)):
    x = torch.randn(M, din, device=dev); w = torch.randn(din, dout, device=dev); b = torch.randn(dout, device=dev)
    y = torch.empty(M, dout, device=dev)
    wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    f = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, 0, ptr(ws), wsb, current_stream()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res["%dx%d" % (din, dout)] = {"ms": round(ms, 4), "TF": round(2.0 * M * din * dout / ms / 1e9, 1)}
print(json.dumps(res))
