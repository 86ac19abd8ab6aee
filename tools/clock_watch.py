#!/usr/bin/env python3
"""Development tool (GPU box): runs one kernel family in a loop for a few seconds while sampling the shader clock and
power with rocm-smi, to tell pipeline limits from power / clock limits.  usage: clock_watch.py gemm|fused|copy"""
import subprocess, sys, threading, time, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dev = torch.device("cuda:0")
if which == "gemm":
    m, d = 1_000_000, 256
    x = torch.randn(m, d, device=dev); w = torch.randn(d, d, device=dev) * 0.05; y = torch.empty(m, d, device=dev)
    fn = lambda: check(lib.kgcn_dense_fwd_f32(ptr(x), m, d, d, ptr(w), d, 0, 0, ptr(y), d, d, current_stream()))
elif which == "copy":
    a = torch.empty(256 << 20, device=dev); b = torch.empty_like(a)
    fn = lambda: b.copy_(a)
else:
    from bench import make_cfg2
    from kgcn_amd import ops
    wl = make_cfg2(100000, dev)
    x, wgt, bias, csr, g = wl["x"].requires_grad_(True), wl["w"].requires_grad_(True), wl["bias"].requires_grad_(True), wl["csr"], wl["g"]
    def fn():
        x.grad = None
        ops.graphconv_fused(x, wgt, bias, csr).backward(g)
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "fclk" in l or "mclk" in l]
            samples.append(" | ".join(x.split(":", 1)[-1].strip()[-40:] for x in s))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.3)
for _ in range(5): fn()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
el = time.time() - t0
stop = True; th.join()
print("%s: %.3f ms per call over %.1f s" % (which, el / n * 1e3, el))
for s in samples[:12]: print("  ", s)
