"""debug: which rows of d inputs are wrong in ops.dense_gather at > 512 tiles"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kgcn_amd import ops
T, N, din, dout = int(sys.argv[1]), 10, 256, 256
rng = np.random.default_rng(1)
x = rng.standard_normal((T, N, din)).astype(np.float32)
w = (rng.standard_normal((din, dout)) / 16).astype(np.float32)
b = rng.standard_normal(dout).astype(np.float32)
gp = rng.standard_normal((T, dout)).astype(np.float32)
gy = rng.standard_normal((T, N, dout)).astype(np.float32)
for tee in (True, False):
    tx, tw, tb = (torch.tensor(a, device="cuda").requires_grad_(True) for a in (x, w, b))
    y, pooled = ops.dense_gather(tx, tw, tb, activation="relu")
    loss = (pooled * torch.tensor(gp, device="cuda")).sum() + ((y * torch.tensor(gy, device="cuda")).sum() if tee else 0.0)
    loss.backward()
    pre = x.astype(np.float64).reshape(T * N, din) @ w.astype(np.float64) + b
    yr = np.maximum(pre, 0)
    g = np.repeat(gp.astype(np.float64), N, axis=0) + (gy.reshape(T * N, dout) if tee else 0.0)
    dpre = g * (yr > 0)
    dx = dpre @ w.astype(np.float64).T
    e = np.abs(tx.grad.cpu().numpy().reshape(T * N, din) - dx).max(1)
    bad = np.nonzero(e > 1e-3)[0]
    print("tee", tee, "rows", T * N, "bad rows", bad.size, "first", bad[:8], "last", bad[-8:], "max", e.max())
    if bad.size:
        # contiguous runs
        runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
        print("  runs:", [(int(r[0]), int(r[-1])) for r in runs[:12]], "n runs", len(runs))
    ew = np.abs(tw.grad.cpu().numpy() - x.astype(np.float64).reshape(T * N, din).T @ dpre).max()
    print("  dW err", ew)
