#!/usr/bin/env python3
"""Per-phase cycles of gemm3_fwd_kernel (development tool, GPU box): builds the library with -DKGCN_PROBE into
gpurun_out/ and runs y = x @ W for 1M x 256 x 256."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libkgcn_g3probe.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = [os.path.join(ROOT, "kgcn_amd", "csrc", f) for f in ("misc.hip", "spmm.hip", "dense.hip", "gemm3.hip", "fused.hip", "pack.hip", "gat.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DKGCN_PROBE",
                       "-fno-slp-vectorize", "-ffp-contract=fast", "-o", out] + src)
lib = ctypes.CDLL(out)
lib.kgcn_dense_fwd_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
lib.kgcn_g3_probe_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
m, din, dout = 1_000_000, 256, 256
x = torch.randn(m, din, device=dev); w = torch.randn(din, dout, device=dev) * 0.05; y = torch.empty(m, dout, device=dev)
probe = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
assert lib.kgcn_g3_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    probe.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.kgcn_dense_fwd_f32(x.data_ptr(), m, din, din, w.data_ptr(), dout, 0, None, y.data_ptr(), dout, dout, s) == 0
    e1.record(); torch.cuda.synchronize()
pr = probe.cpu().numpy().reshape(256, 8, 4).astype(np.float64)
iters = (m / 128 / 256) * (din / 32)
print("launch %.1f us; cycles per chunk-step, by wave: setup | mfma+staging | epilogue (amortised) | barrier wait" % (e0.elapsed_time(e1) * 1e3))
for wv in range(8):
    print("  wave %d: %6.0f %6.0f %6.0f %6.0f" % ((wv,) + tuple(pr[:, wv, k].mean() / iters for k in range(4))))
