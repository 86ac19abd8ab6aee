#!/usr/bin/env python3
"""Per-phase cycle breakdown of graphconv_bwd_pairs_kernel, per ROLE (development tool, GPU box only).
The probe library is prebuilt here:  tools/variants.sh build probe "-DKGCN_PROBE [-D...]"  ->  build/variants/libkgcn_probe.so
usage: KGCN_PROBE_LIB=build/variants/libkgcn_probe.so python tools/pairs_probe.py [graphs] [role_bit]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.environ.get("KGCN_PROBE_LIB", os.path.join(ROOT, "build", "variants", "libkgcn_probe.so"))
import kgcn_amd._lib as L
lib = ctypes.CDLL(out)
for name, (res, args) in L.SIGNATURES.items():
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = args
lib.kgcn_probe_set.argtypes = [ctypes.c_void_p]
from bench import make_cfg2
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
role_bit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
wl = make_cfg2(T, dev)
csr = wl["csr"]
x, g, w = wl["x"], wl["g"], wl["w"]
dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty(64, device=dev)
wsb = lib.kgcn_graphconv_bwd_workspace_bytes(T, 64, 64)
wsp = torch.empty(wsb // 4, device=dev)
probe = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
assert lib.kgcn_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
at = csr.transpose().padded4()
for rep in range(4):
    probe.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.kgcn_graphconv_bwd_f32(at.desc(), p(x), p(w), p(g), 64, 64, p(dx), p(dw), p(db), p(wsp), wsb, s)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0
pr = probe.cpu().numpy().reshape(2048, 8).astype(np.float64)
wave = np.arange(2048) % 8
role = (wave >> 2) if role_bit == 2 else (wave & 1)
iters = T / 1024.0
print("%.1f us/launch (probe build); %.1f iterations per pair; cycles per iteration (s_memtime ticks)" % (e0.elapsed_time(e1) * 1e3, iters))
names = {0: ["prologue", "aggregate half of (i+1) -> planes", "land g(i+2) / CSR, request (i+3)", "dX(i) MFMAs + stores", "barrier wait", "tail",
             "wait for role B's flag", "-"],
         1: ["prologue", "dW(i) MFMAs", "split x(i+1), request x(i+2)", "aggregate half of (i+1) -> planes, flag", "barrier wait", "tail", "-", "-"]}
for r in (0, 1):
    sel = pr[role == r]
    tot = sel.sum(1).mean() / iters
    print("role %s: %.0f per iteration" % ("A (dX)" if r == 0 else "B (dW)", tot))
    for k, n in enumerate(names[r]):
        v = sel[:, k].mean() / iters
        print("   %-36s %9.1f  (%4.1f%%)   min %.0f max %.0f over waves" % (n, v, 100 * v / tot, sel[:, k].min() / iters, sel[:, k].max() / iters))
