#!/usr/bin/env python3
"""The wide-layer GEMM family through the C ABI (GPU box): forward with the W table, dX (W^T table), dX with the activation
derivative (+ gathered read-out gradient), weight gradient -- each checked against fp64 on a row sample and timed with HIP
events.  Run once per library to A/B (KGCN_HIP_LIB=build/variants/libkgcn_prev.so = the bf16 x 3 kernels of round 3).
usage: python tools/gemmh_bench.py [--rows 117888,200000] [--shapes 256x256,84x256] [--quick]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="117888,200000")
ap.add_argument("--shapes", default="256x256")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")


def timeit(fn, reps=args.reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return 1e3 * ts[len(ts) // 2]


def relerr(got, ref):
    return float((got.double() - ref).abs().max() / ref.abs().max())


def rowerr(got, ref):
    s = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    return float(((got.double() - ref).abs() / s).max())


out = {}
for M in [int(r) for r in args.rows.split(",")]:
    for shp in args.shapes.split(","):
        din, dout = [int(v) for v in shp.split("x")]
        g = torch.Generator(device=dev); g.manual_seed(M + din)
        x = torch.randn((M, din), device=dev, generator=g)
        w = (torch.rand((din, dout), device=dev, generator=g) - 0.5) * 0.3
        b = torch.randn((dout,), device=dev, generator=g) * 0.1
        dy = torch.randn((M, dout), device=dev, generator=g) * 1e-3
        y = torch.empty((M, dout), device=dev)
        rows = torch.cat([torch.arange(0, 2048, device=dev), torch.arange(M - 333, M, device=dev),
                          torch.randint(0, M, (2048,), device=dev, generator=g)])
        r = {}
        wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
        ws = torch.zeros((max(wsb, 4) // 4,), device=dev)
        for act, name in ((0, "fwd"), (1, "fwd_sigmoid")):
            f = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, act,
                                                        ptr(ws), wsb, current_stream()))
            f(); torch.cuda.synchronize()
            ref = x[rows].double() @ w.double() + b.double()
            if act:
                ref = torch.sigmoid(ref)
            r[name + "_err"] = rowerr(y[rows], ref)
            r[name + "_us"] = timeit(f)
        # dx = dy W^T
        dx = torch.empty((M, din), device=dev)
        wsb2 = lib.kgcn_dense_fwd_workspace_bytes(dout, din)
        ws2 = torch.zeros((max(wsb2, 4) // 4,), device=dev)
        f = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(dy), M, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0, ptr(ws2),
                                                    wsb2, current_stream()))
        f(); torch.cuda.synchronize()
        r["dx_err"] = rowerr(dx[rows], dy[rows].double() @ w.double().t())
        r["dx_us"] = timeit(f)
        # dx with the activation derivative (the layer output a = sigmoid(...) saved by the forward: y holds it)
        dpre = torch.empty((M, dout), device=dev)
        for act, name in ((1, "dx_dact_sigmoid"), (2, "dx_dact_relu")):
            a = y if act == 1 else (y - 0.5)
            f = lambda: check(lib.kgcn_dense_dx_dact_f32(ptr(dy), ptr(a), M, dout, dout, ptr(w), dout, din, ptr(dx), din, act,
                                                         ptr(dpre), ptr(ws2), wsb2, current_stream()))
            f(); torch.cuda.synchronize()
            a64 = a[rows].double()
            d64 = dy[rows].double() * (a64 * (1 - a64) if act == 1 else (a64 > 0).double())
            r[name + "_err"] = rowerr(dx[rows], d64 @ w.double().t())
            r[name + "_dpre_err"] = relerr(dpre[rows], d64)
            r[name + "_us"] = timeit(f)
        # ... and the gathered read-out gradient (10 nodes per graph)
        if M % 10 == 0 and lib.kgcn_dense_dx_dact_gather_supported(M, din, dout):
            pooled = torch.randn((M // 10, dout), device=dev, generator=g) * 1e-3
            for with_g, name in ((True, "dx_gather"), (False, "dx_gather_only")):
                f = lambda: check(lib.kgcn_dense_dx_dact_gather_f32(ptr(dy) if with_g else None, ptr(pooled), dout, 10, ptr(y), M, dout,
                                                                    dout, ptr(w), dout, din, ptr(dx), din, 1, ptr(dpre), ptr(ws2),
                                                                    wsb2, 0, current_stream()))
                f(); torch.cuda.synchronize()
                a64 = y[rows].double()
                g64 = pooled[rows // 10].double() + (dy[rows].double() if with_g else 0)
                d64 = g64 * a64 * (1 - a64)
                r[name + "_err"] = rowerr(dx[rows], d64 @ w.double().t())
                r[name + "_dpre_err"] = relerr(dpre[rows], d64)
                r[name + "_us"] = timeit(f)
        # weight gradient
        dw = torch.empty((din, dout), device=dev)
        db = torch.empty((dout,), device=dev)
        wgb = lib.kgcn_dense_wgrad_workspace_bytes(M, din, dout)
        wgs = torch.empty((wgb // 4,), device=dev)
        f = lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(dy), dout, M, din, dout, ptr(dw), ptr(db), ptr(wgs), wgb,
                                                   current_stream()))
        f(); torch.cuda.synchronize()
        if not args.quick:
            refw = torch.zeros((din, dout), dtype=torch.float64, device=dev)
            for s in range(0, M, 32768):
                refw += x[s:s + 32768].double().t() @ dy[s:s + 32768].double()
            r["wgrad_err"] = relerr(dw, refw)
            r["wgrad_rowerr"] = rowerr(dw, refw)
            r["dbias_err"] = relerr(db, dy.double().sum(0))
        r["wgrad_us"] = timeit(f)
        # a column of x scaled far down / rows of growing magnitude: the online column scales
        xs = x.clone()
        xs[:, 3] *= 1e-9
        xs[M // 2:] *= 64.0
        dz = dy.clone()
        dz *= (torch.rand((M, 1), device=dev, generator=g) < 0.5)          # whole zero rows (padding rows of a padded batch)
        dz[:37] = 0                                                        # ... and at the head of the first row range
        dz *= torch.exp(3.0 * torch.randn((M, 1), device=dev, generator=g))       # heavy-tailed row magnitudes
        f3 = lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(dz), dout, M, din, dout, ptr(dw), ptr(db), ptr(wgs), wgb,
                                                    current_stream()))
        f3(); torch.cuda.synchronize()
        if not args.quick:
            refw = torch.zeros((din, dout), dtype=torch.float64, device=dev)
            for s in range(0, M, 32768):
                refw += x[s:s + 32768].double().t() @ dz[s:s + 32768].double()
            r["wgrad_zero_rows_err"] = relerr(dw, refw)
        r["wgrad_heavy_tail_us"] = timeit(f3)
        f2 = lambda: check(lib.kgcn_dense_wgrad_f32(ptr(xs), din, ptr(dy), dout, M, din, dout, ptr(dw), ptr(db), ptr(wgs), wgb,
                                                    current_stream()))
        f2(); torch.cuda.synchronize()
        if not args.quick:
            refw = torch.zeros((din, dout), dtype=torch.float64, device=dev)
            for s in range(0, M, 32768):
                refw += xs[s:s + 32768].double().t() @ dy[s:s + 32768].double()
            r["wgrad_scaled_rowerr"] = rowerr(dw, refw)
        fl = 2.0 * M * din * dout
        for k in list(r):
            if k.endswith("_us"):
                r[k.replace("_us", "_TF")] = round(fl / r[k] / 1e6, 1)
        out["%d:%dx%d" % (M, din, dout)] = r
        print("%7d %3dx%3d " % (M, din, dout) + "  ".join("%s %.3g" % (k, v) for k, v in r.items() if not k.endswith("_TF")),
              file=sys.stderr)
print(json.dumps(out))
