#!/usr/bin/env python3
"""Kernel-only timing of the wide-layer GEMMs with the W table already split (what a training step sees): forward, dX with the
activation derivative, weight gradient.  For rocprofv3 / variant libraries (KGCN_HIP_LIB).
usage: python tools/gemmh_prof.py [rows] [din] [dout] [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 117888
din = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((M, din), device=dev, generator=g)
w = (torch.rand((din, dout), device=dev, generator=g) - 0.5) * 0.3
b = torch.randn((dout,), device=dev, generator=g) * 0.1
dy = torch.randn((M, dout), device=dev, generator=g) * 1e-3
y = torch.empty((M, dout), device=dev)
dx = torch.empty((M, din), device=dev)
dpre = torch.empty((M, dout), device=dev)
wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
ws = torch.zeros((max(wsb, 4) // 4,), device=dev)
wsb2 = lib.kgcn_dense_fwd_workspace_bytes(dout, din)
ws2 = torch.zeros((max(wsb2, 4) // 4,), device=dev)
check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, 1, ptr(ws), wsb, current_stream()))
check(lib.kgcn_dense_fwd_ws_f32(ptr(dy), M, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0, ptr(ws2), wsb2, current_stream()))
wgb = lib.kgcn_dense_wgrad_workspace_bytes(M, din, dout)
wgs = torch.empty((wgb // 4,), device=dev)
dw = torch.empty((din, dout), device=dev); db = torch.empty((dout,), device=dev)


def timeit(fn, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) for a, c in ev)
    return round(1e3 * ts[len(ts) // 2], 1)


r = {"rows": M, "din": din, "dout": dout, "lib": os.environ.get("KGCN_HIP_LIB", "shipped")}
r["fwd_us"] = timeit(lambda: check(lib.kgcn_dense_fwd_tab_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, 1,
                                                               ptr(ws), wsb, current_stream())))
r["dx_us"] = timeit(lambda: check(lib.kgcn_dense_fwd_tab_f32(ptr(dy), M, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0,
                                                              ptr(ws2), wsb2, current_stream())))
r["dx_dact_us"] = timeit(lambda: check(lib.kgcn_dense_dx_dact_tab_f32(ptr(dy), ptr(y), M, dout, dout, ptr(w), dout, din, ptr(dx), din,
                                                                      1, ptr(dpre), ptr(ws2), wsb2, current_stream())))
r["wgrad_us"] = timeit(lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(dy), dout, M, din, dout, ptr(dw), ptr(db), ptr(wgs),
                                                              wgb, current_stream())))
print(json.dumps(r))
