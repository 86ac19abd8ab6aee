#!/usr/bin/env python3
"""The table-variant GEMM of gemm3.hip at row counts around the launch-shape decisions (GPU box): y = x W + b (256 -> 256)
and dx = (g (.) relu'(a)) W^T through the C ABI, median of 30 event-timed launches.  Run once with the shipped library and
once with a DEV_KNOBS build + KGCN_GEMM3_CUT=0 (whole 256-column blocks only) to see what the column cuts buy:
  python tools/gemm_cut_bench.py > a.json;  KGCN_HIP_LIB=build/variants/libkgcn_dev.so KGCN_GEMM3_CUT=0 python tools/gemm_cut_bench.py > b.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


res = {}
D = 256
for M in (1024, 4457, 8000, 12000, 16384, 20000, 36160, 49152, 117888, 200000):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((M, D), device=dev, generator=g)
    w = torch.randn((D, D), device=dev, generator=g) * 0.1
    b = torch.randn((D,), device=dev, generator=g)
    y = torch.empty((M, D), device=dev)
    wsb = lib.kgcn_dense_fwd_workspace_bytes(D, D)
    ws = torch.empty((max(wsb, 4) // 4,), device=dev)
    fwd = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, D, D, ptr(w), D, 0, ptr(b), ptr(y), D, D, 0, ptr(ws), wsb, current_stream()))
    fwd(); torch.cuda.synchronize()
    ref = x[-300:].double() @ w.double() + b.double()
    err = float((y[-300:].double() - ref).abs().max() / ref.abs().max())
    t = timeit(fwd)
    # backward of an activated layer: dx = (grad * relu'(a)) W^T, dpre written on the way
    a = torch.relu(torch.randn((M, D), device=dev, generator=g))
    grad = torch.randn((M, D), device=dev, generator=g)
    dpre, dx = torch.empty_like(grad), torch.empty((M, D), device=dev)
    bwd = lambda: check(lib.kgcn_dense_dx_dact_f32(ptr(grad), ptr(a), M, D, D, ptr(w), D, D, ptr(dx), D, 2, ptr(dpre), ptr(ws), wsb,
                                                   current_stream()))
    bwd(); torch.cuda.synchronize()
    rp = grad[-300:].double() * (a[-300:] > 0).double()
    rx = rp @ w.double().t()
    errx = float((dx[-300:].double() - rx).abs().max() / rx.abs().max())
    errp = float((dpre[-300:].double() - rp).abs().max())
    tb = timeit(bwd)
    res[str(M)] = {"fwd_us": t * 1e3, "fwd_tf": 2.0 * M * D * D / t / 1e9, "fwd_err": err, "dx_dact_us": tb * 1e3, "dx_err": errx,
                   "dpre_err": errp}
print(json.dumps(res))
