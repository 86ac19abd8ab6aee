#!/usr/bin/env python3
"""Per-phase cycles of gemm4_fwd_kernel (GPU box; library built with -DKGCN_PROBE: VSRC=gemm4 tools/variants.sh build g4probe "-DKGCN_PROBE").
usage: KGCN_HIP_LIB=build/variants/libkgcn_g4probe.so python tools/gemm4_probe.py [rows] [din]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check, LIB_PATH  # noqa: E402

raw = ctypes.CDLL(LIB_PATH)
raw.kgcn_g4_probe_set.argtypes = [ctypes.c_void_p]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
din = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dout = 256
dev = torch.device("cuda:0")
x = torch.randn((M, din), device=dev); w = torch.randn((din, dout), device=dev); y = torch.empty((M, dout), device=dev)
wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
ws = torch.empty((wsb // 4,), device=dev)
probe = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
assert raw.kgcn_g4_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
for _ in range(3):
    check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, None, ptr(y), dout, dout, 0, ptr(ws), wsb, current_stream()))
torch.cuda.synchronize()
p = probe.cpu().numpy().reshape(1024, 4).astype(np.float64)
items = M / 128 * 2 / 1024
print("rows %d din %d: items per wave %.2f; cycles per item: prologue %.0f  k-loop %.0f (%.0f per k-step)  epilogue %.0f" % (
    M, din, items, p[:, 0].mean() / items, p[:, 1].mean() / items, p[:, 1].mean() / items / ((din + 15) // 16), p[:, 2].mean() / items))
