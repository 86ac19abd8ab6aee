#!/usr/bin/env python3
"""Dense-contraction kernels at the widths of BASELINE configs 3-5 (GPU box): y = x W + b through the C ABI with and
without the workspace of the W fragment table (gemm3 with the table vs gemm3 splitting W in the kernel), dx = dy W^T, correctness against fp64, TF/s
(flops 2 m din dout; fp32 matrix peak 157.3 TF, bf16-split ceiling 2500 / 6 = 417 TF).
usage: python tools/gemm_bench.py [rows]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204_800
res = {}


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for din, dout in ((256, 256), (81, 256), (128, 256), (256, 512), (512, 256)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((M, din), device=dev, generator=g)
    w = torch.randn((din, dout), device=dev, generator=g) * 0.1
    b = torch.randn((dout,), device=dev, generator=g)
    y3, y4 = torch.empty((M, dout), device=dev), torch.empty((M, dout), device=dev)
    wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
    ws = torch.empty((max(wsb, 4) // 4,), device=dev)
    f3 = lambda: check(lib.kgcn_dense_fwd_act_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y3), dout, dout, 0, current_stream()))
    f4 = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y4), dout, dout, 0, ptr(ws), wsb, current_stream()))
    f3(); f4(); torch.cuda.synchronize()
    ref = (x[:4096].double() @ w.double() + b.double())
    e3 = float((y3[:4096].double() - ref).abs().max() / ref.abs().max())
    e4 = float((y4[:4096].double() - ref).abs().max() / ref.abs().max())
    e4t = float((y4[-300:].double() - (x[-300:].double() @ w.double() + b.double())).abs().max() / ref.abs().max())
    t3, t4 = timeit(f3), timeit(f4)
    fl = 2.0 * M * din * dout
    # dx = dy @ W^T : trans_w = 1, K = dout, N = din
    dy = torch.randn((M, dout), device=dev, generator=g)
    dx3, dx4 = torch.empty((M, din), device=dev), torch.empty((M, din), device=dev)
    wsb2 = lib.kgcn_dense_fwd_workspace_bytes(dout, din)
    ws2 = torch.empty((max(wsb2, 4) // 4,), device=dev)
    g3 = lambda: check(lib.kgcn_dense_fwd_act_f32(ptr(dy), M, dout, dout, ptr(w), dout, 1, None, ptr(dx3), din, din, 0, current_stream()))
    g4 = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(dy), M, dout, dout, ptr(w), dout, 1, None, ptr(dx4), din, din, 0, ptr(ws2), wsb2, current_stream()))
    g3(); g4(); torch.cuda.synchronize()
    refx = dy[:4096].double() @ w.double().t()
    ex3 = float((dx3[:4096].double() - refx).abs().max() / refx.abs().max())
    ex4 = float((dx4[:4096].double() - refx).abs().max() / refx.abs().max())
    tx3, tx4 = timeit(g3), timeit(g4)
    res["%dx%d" % (din, dout)] = {"rows": M, "fwd_gemm3_ms": t3, "fwd_table_ms": t4, "fwd_gemm3_TF": fl / t3 / 1e9, "fwd_table_TF": fl / t4 / 1e9,
                                  "fwd_err_gemm3": e3, "fwd_err_table": e4, "fwd_err_table_tail": e4t, "ws_bytes": wsb,
                                  "dx_gemm3_ms": tx3, "dx_table_ms": tx4, "dx_gemm3_TF": fl / tx3 / 1e9, "dx_table_TF": fl / tx4 / 1e9,
                                  "dx_err_gemm3": ex3, "dx_err_table": ex4, "dx_ws_bytes": wsb2}
print(json.dumps(res, indent=1))
