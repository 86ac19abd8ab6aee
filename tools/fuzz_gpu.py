#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: many random shapes per kernel family against the oracle (checker only).
Not part of the pytest suites (minutes of oracle time); run to hunt boundary bugs.  usage: fuzz_gpu.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import kgcn_oracle as K  # noqa: E402
from kgcn_amd import BatchedAdjacency, BatchedCSR, layers, ops  # noqa: E402

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
fails = []


def check(name, got, ref, rel=2e-5, atol=2e-5, ctx=None):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64).reshape(got.shape)
    tol = atol + rel * (np.abs(ref).max() if ref.size else 0.0)
    err = np.abs(got - ref).max() if ref.size else 0.0
    if not np.isfinite(got).all() or err > tol:
        fails.append((name, ctx, float(err), float(tol)))


def rand_graphs(T, N, density, empty_every=0, dup=False):
    adjs = []
    for t in range(T):
        if empty_every and t % empty_every == empty_every - 1:
            adjs.append([(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [N, N])])
            continue
        dense = (rng.random((N, N)) < density) * rng.standard_normal((N, N))
        idx = np.argwhere(dense != 0).astype(np.int32)
        val = dense[dense != 0].astype(np.float32)
        if dup and len(idx):
            k = rng.integers(0, len(idx), size=max(1, len(idx) // 5))
            idx, val = np.concatenate([idx, idx[k]]), np.concatenate([val, val[k]])
            p = rng.permutation(len(idx))
            idx, val = idx[p], val[p]
        adjs.append([(idx, val, [N, N])])
    return adjs


t_start = time.time()
for case in range(CASES):
    # ---- fused / unfused GraphConv, any N <= 32, widths <= 64 ------------------------------------------
    N, din, dout, T = int(rng.integers(1, 33)), int(rng.integers(1, 65)), int(rng.integers(1, 65)), int(rng.integers(1, 90))
    dup = bool(rng.integers(0, 2))
    adjs = rand_graphs(T, N, rng.uniform(0.05, 0.6), empty_every=int(rng.integers(0, 6)), dup=dup)
    x = rng.standard_normal((T, N, din)).astype(np.float32)
    w = K.glorot_uniform(rng, din, dout); b = rng.standard_normal((1, dout)).astype(np.float32)
    g = rng.standard_normal((T, N, dout)).astype(np.float32)
    csr = BatchedCSR.from_coo_list([a[0] for a in adjs], rows=N, cols=N, device=dev)
    ctx = ("graphconv", N, din, dout, T)
    tx, tw, tb = t32(x).requires_grad_(True), t32(w).requires_grad_(True), t32(b).requires_grad_(True)
    if ops.graphconv_fused_supported(csr, din, dout):
        out = ops.graphconv_fused(tx, tw, tb, csr)
    else:
        out = ops.bspmm(csr, ops.dense(tx.reshape(T * N, din), tw, tb)).reshape(T, N, dout)
    out.backward(t32(g))
    check("conv fwd", out, K.graphconv_fwd_fast(x, adjs, [w], [b]), ctx=ctx)
    dx, dw, db = K.graphconv_bwd_fast(x, adjs, [w], g)
    check("conv dx", tx.grad, dx, ctx=ctx); check("conv dw", tw.grad, dw[0], ctx=ctx); check("conv db", tb.grad, db[0], ctx=ctx)
    # ---- rectangular SpMM, any D --------------------------------------------------------------------------
    D = int(rng.integers(1, 200))
    rhs = rng.standard_normal((T, N, D)).astype(np.float32)
    check("bspmm", ops.bspmm(csr, t32(rhs)), np.stack(K.bspmm([a[0] for a in adjs], list(rhs))), ctx=("bspmm", N, D, T))
    check("bspmm^T", ops.bspmm(csr.transpose(), t32(rhs)),
          np.stack(K.bspmm([a[0] for a in adjs], list(rhs), adjoint_a=True)), ctx=("bspmmT", N, D, T))
    # ---- max pooling and GAT ---------------------------------------------------------------------------------
    Dm = int(rng.integers(1, 70))
    xm = rng.standard_normal((T, N, Dm)).astype(np.float32)
    adj = BatchedAdjacency([csr])
    txm = t32(xm).requires_grad_(True)
    gm = rng.standard_normal(xm.shape).astype(np.float32)
    if not dup:      # tf.sparse_tensor_to_dense rejects repeated indices: max pooling is defined for unique entries only
        om = ops.graph_maxpool(txm, adj)
        om.backward(t32(gm))
        check("maxpool fwd", om, K.graph_maxpool_fwd(xm, adjs), ctx=("maxpool", N, Dm, T))
        check("maxpool bwd", txm.grad, K.graph_maxpool_bwd(xm, adjs, gm), ctx=("maxpool", N, Dm, T))
    wa = [(rng.standard_normal((2 * Dm, 1)) * 0.3).astype(np.float32)]
    txg, twa = t32(xm * 0.5).requires_grad_(True), t32(wa[0]).requires_grad_(True)
    og = ops.gat(txg, adj, [twa])
    og.backward(t32(gm))
    check("gat fwd", og, K.gat_fwd(xm * 0.5, adjs, wa), ctx=("gat", N, Dm, T))
    dxg, dwa = K.gat_bwd(xm * 0.5, adjs, wa, gm)
    check("gat dx", txg.grad, dxg, rel=1e-4, ctx=("gat", N, Dm, T)); check("gat dwa", twa.grad, dwa[0], rel=1e-4, ctx=("gat", N, Dm, T))
    # ---- decoders ------------------------------------------------------------------------------------------------
    wv = rng.standard_normal(Dm).astype(np.float32)
    txd, twd = t32(xm).requires_grad_(True), t32(wv).requires_grad_(True)
    od = ops.gram(txd, twd)
    gd = rng.standard_normal((T, N, N)).astype(np.float32)
    od.backward(t32(gd))
    check("gram fwd", od, K.gram_fwd(xm, wv), ctx=("gram", N, Dm, T))
    dxd, dwd = K.gram_bwd(xm, wv, gd)
    check("gram dx", txd.grad, dxd, ctx=("gram", N, Dm, T)); check("gram dw", twd.grad, dwd, rel=1e-4, ctx=("gram", N, Dm, T))
    # ---- dense, both kernel families (gemm3 above 128 output columns) ------------------------------------------------
    M, di, do = int(rng.integers(1, 3000)), int(rng.integers(1, 400)), int(rng.integers(1, 400))
    xd = rng.standard_normal((M, di)).astype(np.float32)
    wd = K.glorot_uniform(rng, di, do); bd = rng.standard_normal(do).astype(np.float32)
    gdn = rng.standard_normal((M, do)).astype(np.float32)
    tx2, tw2, tb2 = t32(xd).requires_grad_(True), t32(wd).requires_grad_(True), t32(bd).requires_grad_(True)
    y = ops.dense(tx2, tw2, tb2)
    y.backward(t32(gdn))
    x64, w64, g64 = xd.astype(np.float64), wd.astype(np.float64), gdn.astype(np.float64)
    ctx = ("dense", M, di, do)
    check("dense fwd", y, x64 @ w64 + bd, ctx=ctx); check("dense dx", tx2.grad, g64 @ w64.T, ctx=ctx)
    check("dense dw", tw2.grad, x64.T @ g64, ctx=ctx); check("dense db", tb2.grad, g64.sum(0), ctx=ctx)
    # ---- device-side batch assembly ----------------------------------------------------------------------------------
    sel = rng.integers(-1, T, size=int(rng.integers(0, 3 * T + 2)))
    got = csr.gather(sel)
    host = BatchedCSR.from_coo_list([adjs[s][0] if s >= 0 else (np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [N, N])
                                     for s in sel], rows=N, cols=N, device=dev) if len(sel) else None
    if host is not None and not (torch.equal(got.rowptr, host.rowptr) and torch.equal(got.cv, host.cv)):
        fails.append(("gather", (N, T, len(sel)), 0, 0))
print("%d cases in %.1f s, %d failures" % (CASES, time.time() - t_start, len(fails)))
for f in fails[:30]:
    print("  FAIL", f)
sys.exit(1 if fails else 0)
