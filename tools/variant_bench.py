#!/usr/bin/env python3
"""Correctness + timing of the fused GraphConv kernels of ONE build of the library (development tool, GPU box).
The library is chosen with KGCN_HIP_LIB (kgcn_amd/_lib.py); tools/variants.sh runs this once per variant.
Checks fwd/bwd of a 5,003-graph cfg2 batch against oracle/kgcn_ref.c, then times fwd and bwd at 100k graphs
(median / p10 / p90 of 40 launches, HIP events)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_cfg2, algorithmic_bytes  # noqa: E402
from kgcn_amd._lib import lib, ptr, current_stream, check, LIB_PATH  # noqa: E402
from oracle import ref_c  # noqa: E402

dev = torch.device("cuda:0")
D, N = 64, 32
res = {"lib": os.path.relpath(LIB_PATH, ROOT)}


def run(wl, T):
    csr, x, g, w, b = wl["csr"], wl["x"], wl["g"], wl["w"], wl["bias"].reshape(-1)
    p4, p4t = csr.padded4(), csr.transpose().padded4()
    out, dx = torch.empty_like(x), torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty(D, device=dev)
    wsb = lib.kgcn_graphconv_bwd_workspace_bytes(T, D, D)
    wsp = torch.empty(wsb // 4, device=dev)
    fwd = lambda: check(lib.kgcn_graphconv_fwd_f32(p4.desc(), ptr(x), ptr(w), ptr(b), D, D, ptr(out), current_stream()))
    bwd = lambda: check(lib.kgcn_graphconv_bwd_f32(p4t.desc(), ptr(x), ptr(w), ptr(g), D, D, ptr(dx), ptr(dw), ptr(db),
                                                   ptr(wsp), wsb, current_stream()))
    return fwd, bwd, out, dx, dw, db


for Tc in (() if os.environ.get('VB_TIMING_ONLY') else (5003, 1, 1021)):
    wl = make_cfg2(Tc, dev, normalize=True)
    wl["bias"] = torch.randn(1, D, device=dev) * 0.1
    fwd, bwd, out, dx, dw, db = run(wl, Tc)
    dx.fill_(float("nan"))
    fwd(); bwd()
    torch.cuda.synchronize()
    off = wl["off"]
    xh, gh, wh, bh = (wl[k].cpu().numpy() for k in ("x", "g", "w", "bias"))
    ro = ref_c.graphconv_fwd(off, wl["idx"], wl["val"], xh, wh, bh)
    rdx, rdw, rdb = ref_c.graphconv_bwd(off, wl["idx"], wl["val"], xh, wh, gh)
    rel = lambda a, r: float(np.abs(a.cpu().numpy().reshape(r.shape) - r).max() / max(1.0, np.abs(r).max()))
    res["err_T%d" % Tc] = dict(out=rel(out, ro), dx=rel(dx, rdx), dw=rel(dw, rdw), db=rel(db, rdb))

T = 100_000
wl = make_cfg2(T, dev)
fwd, bwd, *_ = run(wl, T)
ab = algorithmic_bytes(N, D, D, wl["nnz_per_graph"])


def timeit(fn, reps=40):
    for _ in range(25):
        fwd(); bwd()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b_ in ev:
        a.record(); fn(); b_.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) for a, b_ in ev)
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[(9 * len(ts)) // 10]


# alone, and interleaved as in the benchmark step (the clocks the pair sustains differ from either alone)
f, b_ = timeit(fwd), timeit(bwd)
res["fwd_ms"], res["bwd_ms"] = f, b_
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(40)]
for e in ev:
    e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
torch.cuda.synchronize()
fm = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
bm = float(np.median([e[1].elapsed_time(e[2]) for e in ev]))
res["pair_fwd_ms"], res["pair_bwd_ms"] = fm, bm
res["pair_Mgraphs_s"] = T / ((fm + bm) * 1e-3) / 1e6
res["bwd_frac"] = ab["bwd"] * T / (bm * 1e-3) / 8e12
res["fwd_frac"] = ab["fwd"] * T / (fm * 1e-3) / 8e12
print(json.dumps(res))
