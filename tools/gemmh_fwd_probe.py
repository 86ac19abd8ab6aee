#!/usr/bin/env python3
"""Per-phase cycles of gemmh_fwd_kernel<0, 16> (development tool; as tools/gemmb_probe.py: --build cross-compiles gemmh.hip with
-DKGCN_PROBE into build/libkgcn_ghfprobe.so).  usage: python tools/gemmh_fwd_probe.py --build | python tools/gemmh_fwd_probe.py [rows]"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "build", "libkgcn_ghfprobe.so")
if "--build" in sys.argv:
    cs = os.path.join(ROOT, "kgcn_amd", "csrc")
    obj = "/tmp/gemmh_probe.o"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast"]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DKGCN_PROBE", "-c", os.path.join(cs, "gemmh.hip"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(cs, "*.o"))) if not o.endswith("gemmh.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others)
    print("built", out); sys.exit(0)
import numpy as np
import torch
os.environ["KGCN_HIP_LIB"] = out
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check      # noqa: E402
plib = ctypes.CDLL(out); plib.kgcn_gh_probe_set.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
x = torch.randn((M, 256), device=dev); w = torch.randn((256, 256), device=dev) * 0.06; b = torch.zeros(256, device=dev)
y = torch.empty((M, 256), device=dev)
tb = int(lib.kgcn_dense_fwd_workspace_bytes(256, 256)); tab = torch.empty((tb // 4,), device=dev)
probe = torch.zeros(2 * 256 * 8 * 8, dtype=torch.int64, device=dev)
assert plib.kgcn_gh_probe_set(ctypes.c_void_p(probe.data_ptr())) == 0
f = lambda: check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, 256, 256, ptr(w), 256, 0, ptr(b), ptr(y), 256, 256, 2, ptr(tab), tb, current_stream()))
for _ in range(3):
    f()
torch.cuda.synchronize(); probe.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
pr = probe.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
pr = pr[pr.sum(axis=(1, 2)) > 0]
tiles = -(-M // 64) / 256
names = ["loop head", "16 k-steps: 96 MFMAs + the next tile's staging", "scale / bias / activation", "32 stores", "barrier"]
print("launch %.1f us (probed, incl. the table split), %.1f tiles per workgroup; cycles per 64-row tile, mean over the waves (min .. max):" % (e0.elapsed_time(e1) * 1e3, tiles))
tot = 0
for k, n in enumerate(names):
    c = pr[:, :, k] / tiles
    tot += c.mean()
    print("  %-50s %7.0f   (%6.0f .. %6.0f)" % (n, c.mean(), c.min(), c.max()))
print("  %-50s %7.0f" % ("sum", tot))
