#!/usr/bin/env python3
"""Per-phase cycles of gemmb_kernel (development tool, GPU box).  The probe library is cross-compiled in the build container
(tools/gemmb_probe.py --build: gemmb.hip with -DKGCN_PROBE, linked with the shipped objects into build/libkgcn_gbprobe.so, which
travels with the gpurun snapshot); on the GPU box it runs the one-pass backward of a relu layer [rows x 256] -> 256.  Every probe
point reads the cycle counter through s_memtime, which returns through lgkmcnt -- it also waits for the wave's outstanding LDS
operations: the phases are what a wave WAITS for, and the probed kernel runs slower than the shipped one.
usage: python tools/gemmb_probe.py --build | python tools/gemmb_probe.py [rows] [act: none|relu|sigmoid]"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "build", "libkgcn_gbprobe.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cs = os.path.join(ROOT, "kgcn_amd", "csrc")
    obj = "/tmp/gemmb_probe.o"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast"]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DKGCN_PROBE", "-c", os.path.join(cs, "gemmb.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(cs, "*.o"))) if not o.endswith("gemmb.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
    print("built", out)
    sys.exit(0)
import numpy as np
import torch
os.environ["KGCN_HIP_LIB"] = out
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check      # noqa: E402
plib = ctypes.CDLL(out)
plib.kgcn_gb_probe_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(args[0]) if args else 200000
act = {"none": 0, "sigmoid": 1, "relu": 2}[args[1] if len(args) > 1 else "relu"]
x = torch.randn((M, 256), device=dev); g = torch.randn((M, 256), device=dev); a = torch.randn((M, 256), device=dev)
w = torch.randn((256, 256), device=dev) * 0.06
dx = torch.empty((M, 256), device=dev); dw = torch.empty((256, 256), device=dev); db = torch.empty((256,), device=dev)
tb = int(lib.kgcn_dense_fwd_workspace_bytes(256, 256)); tab = torch.empty((tb // 4,), device=dev)
wsb = int(lib.kgcn_dense_wgrad_workspace_bytes(M, 256, 256)); ws = torch.empty((wsb // 4,), device=dev)
probe = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
assert plib.kgcn_gb_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
f = lambda ready: check(lib.kgcn_dense_bwd_f32(ptr(g), None, 0, 0, ptr(a) if act else None, act, 256, ptr(x), 256, M, 256, 256, ptr(w), 256,
                                              ptr(dx), 256, ptr(dw), ptr(db), ptr(tab), tb, ready, ptr(ws), wsb, current_stream()))
f(0)
for _ in range(3):
    f(1)
torch.cuda.synchronize()
probe.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(1); e1.record(); torch.cuda.synchronize()
pr = probe.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
stages = -(-(-(-M // 32)) // 128)
print("launch %.1f us (probed, incl. second stage), %.1f stages per workgroup; cycles per stage, mean over the role's waves (min .. max wave):"
      % (e0.elapsed_time(e1) * 1e3, stages))
roles = (("dX role (waves 0-3: staging, dX)", pr[:, :4, :],
          ["-", "16 k-steps: 48 MFMAs + the next stage's staging", "dX epilogue + stores", "-", "-", "barrier"]),
         ("dW role (waves 4-7)", pr[:, 4:, :],
          ["x'' split (waits: x fragments, row exponents)", "16 tiles: 48 MFMAs (waits: LDS fragments)", "-", "-", "-", "barrier (incl. loop)"]))
for title, v, names in roles:
    print(" ", title)
    tot = 0
    for k, n in enumerate(names):
        if n == "-":
            continue
        c = v[:, :, k] / stages
        tot += c.mean()
        print("    %-52s %7.0f   (%6.0f .. %6.0f)" % (n, c.mean(), c.min(), c.max()))
    print("    %-52s %7.0f" % ("sum", tot))
