#!/usr/bin/env python3
"""Randomised shapes through the f16 two-piece GEMM family (rows >= 16,384; widths that are not multiples of 16 / 32 / 64; ragged
last tiles and stages) against fp64: forward with bias + activation, dX, dX with act', weight gradient + dbias.
usage: python tools/fuzz_gemmh.py [cases] [seed]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import ops      # noqa: E402
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
fails = 0
for c in range(cases):
    m = int(rng.integers(16384, 40000))
    din = int(rng.choice([100, 132, 160, 200, 228, 252, 256, 64, 96]))
    dout = int(rng.choice([132, 136, 160, 200, 250, 256, 300, 384]))
    act = [None, "relu", "sigmoid", "tanh"][int(rng.integers(0, 4))]
    # (a saturating activation turns the pre-activation's fp32-class error, ~1e-7 sum|x w|, into an ABSOLUTE error of its O(1) output:
    # large inputs only for the linear cases)
    x = torch.randn((m, din), device=dev) * float(10 ** rng.uniform(-3, 3 if act in (None, "relu") else 0.3))
    w = (torch.rand((din, dout), device=dev) - 0.5) * 0.2
    b = torch.randn((dout,), device=dev) * 0.1
    gy = torch.randn((m, dout), device=dev) * 1e-2
    tx, tw, tb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.dense(tx, tw, tb, activation=act)
    (y * gy).sum().backward()
    rows = torch.cat([torch.arange(0, 300, device=dev), torch.arange(m - 300, m, device=dev), torch.randint(0, m, (600,), device=dev)])
    x64, w64 = x.double(), w.double()
    pre = x64[rows] @ w64 + b.double()
    f = {None: lambda v: v, "relu": lambda v: v.clamp_min(0), "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act]
    yr = f(pre)
    ya = y.detach().double()
    df = {None: lambda a: torch.ones_like(a), "relu": lambda a: (a > 0).double(), "sigmoid": lambda a: a * (1 - a), "tanh": lambda a: 1 - a * a}[act]
    dpre_all = gy.double() * df(ya)
    dxr = dpre_all[rows] @ w64.t()
    dwr = torch.zeros((din, dout), dtype=torch.float64, device=dev)
    for s in range(0, m, 8192):
        dwr += x64[s:s + 8192].t() @ dpre_all[s:s + 8192]
    dbr = dpre_all.sum(0)
    rel = lambda got, ref: float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300))
    e = {"y": rel(y.detach()[rows], yr), "dx": rel(tx.grad[rows], dxr), "dw": rel(tw.grad, dwr), "db": rel(tb.grad, dbr)}
    bad = {k: v for k, v in e.items() if not (v < 5e-6)}
    if bad:
        fails += 1
    print("%2d m=%d %d->%d %-7s %s%s" % (c, m, din, dout, act, " ".join("%s %.1e" % kv for kv in e.items()), "   <-- FAIL" if bad else ""))
print("%d cases, %d failures" % (cases, fails))
