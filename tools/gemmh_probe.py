#!/usr/bin/env python3
"""Per-phase cycles of gemmh_wgradl_kernel (development tool, GPU box): builds the library with -DKGCN_PROBE into gpurun_out/ and runs
dW = x^T dy for [rows x 256] x [rows x 256].  Every probe point reads s_memtime, which returns through lgkmcnt -- i.e. it also waits for
the wave's outstanding LDS operations: the phases are what a wave WAITS for, the kernel itself runs slower than the shipped one.
usage: python tools/gemmh_probe.py [rows]"""
import ctypes, glob, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "libkgcn_ghprobe.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = sorted(glob.glob(os.path.join(ROOT, "kgcn_amd", "csrc", "*.hip")))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DKGCN_PROBE",
                       "-Wno-unused-function", "-ffp-contract=fast", "-o", out] + src)
os.environ["KGCN_HIP_LIB"] = out
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check      # noqa: E402
plib = ctypes.CDLL(out)
plib.kgcn_gh_probe_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 117888
din = dout = 256
x = torch.randn((M, din), device=dev); dy = torch.randn((M, dout), device=dev) * 1e-3
dw = torch.empty((din, dout), device=dev); db = torch.empty((dout,), device=dev)
wgb = lib.kgcn_dense_wgrad_workspace_bytes(M, din, dout)
wgs = torch.empty((wgb // 4,), device=dev)
probe = torch.zeros(2 * 256 * 8 * 8, dtype=torch.int64, device=dev)
assert plib.kgcn_gh_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
f = lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(dy), dout, M, din, dout, ptr(dw), ptr(db), ptr(wgs), wgb, current_stream()))
for _ in range(3):
    f()
torch.cuda.synchronize()
probe.zero_()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); f(); b.record(); torch.cuda.synchronize()
pr = probe.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
pr = pr[pr.sum(axis=(1, 2)) > 0]
stages = M / 32 / 128
names = ["issue loads", "split_check (raw data)", "mult_check + frags 0", "k-step 0: MFMAs + emit dy", "frags 1", "k-step 1: MFMAs + emit x",
         "barrier", "loop / rest"]
print("launch %.1f us (probed), %d workgroups, %.1f stages each; cycles per stage and wave:" % (a.elapsed_time(b) * 1e3, pr.shape[0], stages))
for k, n in enumerate(names):
    print("  %-28s light waves 0-3: %7.0f   heavy waves 4-7: %7.0f" % (n, pr[:, :4, k].mean() / stages, pr[:, 4:, k].mean() / stages))
print("  %-28s light waves 0-3: %7.0f   heavy waves 4-7: %7.0f" % ("sum", pr[:, :4, :].sum(axis=2).mean() / stages, pr[:, 4:, :].sum(axis=2).mean() / stages))
