#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused kernels (development tool, GPU box only).
Builds kgcn_amd/csrc with -DKGCN_PROBE into gpurun_out/libkgcn_probe.so, runs one forward and
one backward launch on the cfg2 workload and prints the per-graph average s_memtime cycles per phase."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libkgcn_probe.so")
src = [os.path.join(ROOT, "kgcn_amd", "csrc", f) for f in ("misc.hip", "spmm.hip", "dense.hip", "gemm3.hip", "wtable.hip", "gemmn.hip", "wgradn.hip", "narrow.hip", "fused.hip", "pack.hip", "gat.hip", "bn.hip")]
if os.environ.get("KGCN_PROBE_LIB"):      # prebuilt (tools/variants.sh build probe "-DKGCN_PROBE")
    out = os.environ["KGCN_PROBE_LIB"]
else:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                         "-DKGCN_PROBE", "-fno-slp-vectorize", "-o", out] + [a for a in sys.argv[1:] if a.startswith("-D")] + src)
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
import kgcn_amd._lib as L
L.LIB_PATH = out
lib = ctypes.CDLL(out)
for name, (res, args) in L.SIGNATURES.items():
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = args
lib.kgcn_probe_set.argtypes = [ctypes.c_void_p]
from bench import make_cfg2
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
wl = make_cfg2(T, dev)
csr = wl["csr"]
x, g, w, b = wl["x"], wl["g"], wl["w"], wl["bias"].reshape(-1)
outt = torch.empty_like(x)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
db = torch.empty(64, device=dev)
wsb = lib.kgcn_graphconv_bwd_workspace_bytes(T, 64, 64)
wsp = torch.empty(wsb // 4, device=dev)
probe = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
NW = {'fwd': 2048.0, 'bwd': 1024.0}
assert lib.kgcn_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
names_f = ["loop-top", "land x + issue x(next)", "aggregate(t-1) + store", "land CSR + issue + split + mfma(t)", "FW->LDS"]
names_g = ["loop top", "land g + CSR", "aggregate -> dFW", "land x + issue prefetch", "dW (split + bf16 mfma)", "dX mfma",
           "dX -> LDS -> HBM"]
names_b = ["prologue+iter0", "A: k-step 1 MFMAs + trailing halves + emits", "B: MFMA 0-11", "B: MFMA 12-23", "B: MFMA 24-35",
           "B: MFMA 36-47", "A: k-step 0, 24 MFMA slots", "A: k-step 0 trailing halves"]
for which in ("fwd", "bwd"):
    for rep in range(3):
        probe.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if which == "fwd":
            rc = lib.kgcn_graphconv_fwd_f32(csr.padded4().desc(), p(x), p(w), p(b), 64, 64, p(outt), s)
        else:
            rc = lib.kgcn_graphconv_bwd_f32(csr.transpose().padded4().desc(), p(x), p(w), p(g), 64, 64, p(dx), p(dw), p(db),
                                            p(wsp), wsb, s)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
    pr = probe.cpu().numpy().reshape(2048, 8).astype(np.float64)
    graphs_per_wave = T / NW[which]
    nw = int((pr.sum(1) > 0).sum())                     # waves that ran
    graphs_per_wave = T / nw
    pr = pr[:nw]
    tot = pr.sum(1).mean() / graphs_per_wave
    print("%s: %.1f us/launch (probe build), per graph per wave: %.0f cycles (s_memtime ticks @100MHz? see ratio)"
          % (which, e0.elapsed_time(e1) * 1e3, tot))
    names = names_f if which == "fwd" else names_b
    print("   waves: %d" % nw)
    for k, n in enumerate(names):
        v = pr[:, k].mean() / graphs_per_wave
        print("   %-28s %9.1f  (%4.1f%%)" % (n, v, 100 * v / tot))
