#!/usr/bin/env python3
"""Measured error of the HIP GraphConv paths against the fp64 oracle (GPU box).  The fused kernels run
their contractions on the bf16 matrix pipe with an exact 3-way bf16 split of every fp32 operand (six
products, fp32 accumulate); the unfused kernels (kgcn_dense_*) use v_mfma_f32_32x32x2_f32.  Both are
compared here with a plain fp32 evaluation (numpy float32) of the same formulas, so the numbers show
where each sits relative to ordinary fp32 rounding.  Prints one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kgcn_oracle as K  # noqa: E402  (checker only)
from kgcn_amd import BatchedCSR, ops  # noqa: E402

rng = np.random.default_rng(7)
T, N, D = 512, 32, 64
adjs = K.synth_mol_graphs(rng, T, N, 3, normalize=True)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
x = (rng.standard_normal((T, N, D)) * scale).astype(np.float32)
w = K.glorot_uniform(rng, D, D)
b = rng.standard_normal((1, D)).astype(np.float32)
g = rng.standard_normal((T, N, D)).astype(np.float32)
ref = K.graphconv_fwd_fast(x, adjs, [w], [b])
dx64, dw64, db64 = K.graphconv_bwd_fast(x, adjs, [w], g)
f32 = K.graphconv_fwd_fast(x, adjs, [w], [b], dtype=np.float32)
dx32, dw32, db32 = K.graphconv_bwd_fast(x, adjs, [w], g, dtype=np.float32)
dev = torch.device("cuda:0")
csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def run(fused):
    tx, tw, tb = t(x).requires_grad_(True), t(w).requires_grad_(True), t(b).requires_grad_(True)
    if fused:
        out = ops.graphconv_fused(tx, tw, tb, csr)
    else:
        out = ops.bspmm(csr, ops.dense(tx.reshape(T * N, D), tw, tb)).reshape(T, N, D)
    out.backward(t(g))
    return [v.detach().cpu().numpy().astype(np.float64) for v in (out, tx.grad, tw.grad, tb.grad)]


def err(a, r):
    return {"max_abs": float(np.abs(a - r).max()), "max_abs_over_max_ref": float(np.abs(a - r).max() / np.abs(r).max())}


res = {"graphs": T, "x_scale": scale}
for name, vals in (("fused (bf16x3 split MFMA)", run(True)), ("unfused (f32 MFMA)", run(False)),
                   ("numpy float32", [f32, dx32, dw32[0], db32[0]])):
    res[name] = {k: err(np.asarray(v, np.float64).reshape(np.asarray(r).shape), np.asarray(r))
                 for k, v, r in zip(("out", "dX", "dW", "dbias"), vals, (ref, dx64, dw64[0], db64[0]))}
print(json.dumps(res, indent=1))
