import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from kgcn_amd._lib import lib, ptr, current_stream, check
dev = torch.device("cuda:0")
M, din, dout = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 256, 256
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((M, din), device=dev, generator=g)
w = (torch.rand((din, dout), device=dev, generator=g) - 0.5) * 0.3
y = torch.zeros((M, dout), device=dev)
wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout); ws = torch.zeros((wsb // 4,), device=dev)
check(lib.kgcn_dense_fwd_ws_f32(ptr(x), M, din, din, ptr(w), dout, 0, None, ptr(y), dout, dout, 0, ptr(ws), wsb, current_stream()))
torch.cuda.synchronize()
ref = (x.double() @ w.double()).cpu().numpy(); got = y.cpu().numpy()
err = np.abs(got - ref)
print("max err", err.max(), "ref max", np.abs(ref).max())
e = err.reshape(M // 64, 64, dout // 32, 32)
bt = e.max(axis=(1, 2, 3)); print("tiles bad:", np.nonzero(bt > 1e-3)[0][:40], "of", len(bt))
print("by row in tile (first tile):", e[0].max(axis=(1, 2)).round(2))
print("by col block:", e.max(axis=(0, 1, 3)).round(3))
print("by col in block:", e.max(axis=(0, 1, 2)).round(2))
r = got[:8, :4] / ref[:8, :4]
print("ratio got/ref:\n", r.round(3))
# is got a row permutation of ref?
for rr in range(4):
    d = np.abs(ref[:64] - got[rr][None, :]).max(axis=1)
    print("got row", rr, "closest ref row", d.argmin(), d.min())
