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
    # ---- multi-channel GraphConv (round 6: channel-loop Bconv, fan-out adjoint, both routes of the layer), C = 2 .. 9 ------------
    Cm, Nm = int(rng.integers(2, 10)), int(rng.integers(1, 70))
    dim_, dom = int(rng.integers(1, 40)), int(rng.integers(1, 70))
    Tm = int(rng.integers(1, 60)) if case % 3 else int(rng.integers(1100 // max(Nm, 1) + 1, 1100 // max(Nm, 1) + 40))
    adjm = [[rand_graphs(1, Nm, rng.uniform(0.02, 0.4), dup=bool(rng.integers(0, 2)))[0][0] for _ in range(Cm)] for _ in range(Tm)]
    xmc = rng.standard_normal((Tm, Nm, dim_)).astype(np.float32)
    gmc = rng.standard_normal((Tm, Nm, dom)).astype(np.float32)
    lay = layers.GraphConv(dom, Cm).to(dev)
    lay.build((Tm, Nm, dim_), dev)
    with torch.no_grad():
        for bb in lay.bias:
            bb.copy_(t32(rng.standard_normal((1, dom)) * 0.1))
    txc = t32(xmc).requires_grad_(True)
    oc = lay(txc, adj=BatchedAdjacency.from_adjs(adjm, n_nodes=Nm, device=dev))
    oc.backward(t32(gmc))
    wl_, bl_ = [p_.detach().cpu().numpy() for p_ in lay.w], [p_.detach().cpu().numpy() for p_ in lay.bias]
    ctx = ("multi-channel conv", Cm, Nm, dim_, dom, Tm)
    check("mc fwd", oc, K.graphconv_fwd(xmc, adjm, wl_, bl_), ctx=ctx)
    dxm, dwm, dbm = K.graphconv_bwd(xmc, adjm, wl_, bl_, gmc)
    check("mc dx", txc.grad, dxm, ctx=ctx)
    for cc in range(Cm):
        check("mc dw%d" % cc, lay.w[cc].grad, dwm[cc], ctx=ctx); check("mc db%d" % cc, lay.bias[cc].grad, dbm[cc].reshape(1, dom), ctx=ctx)
    # ---- the FULL-shape backward with two waves per graph slot (>= 2,048 graphs) ---------------------------------------------------
    if case % 8 == 0:
        Tf = int(rng.integers(2048, 2700))
        adjf = K.synth_mol_graphs(rng, Tf, 32, int(rng.integers(0, 6)), normalize=bool(rng.integers(0, 2)))
        for tz in rng.integers(0, Tf, size=5):
            adjf[int(tz)] = [(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [32, 32])]        # empty graphs
        xf = rng.standard_normal((Tf, 32, 64)).astype(np.float32); gf = rng.standard_normal((Tf, 32, 64)).astype(np.float32)
        wf = K.glorot_uniform(rng, 64, 64); bf = rng.standard_normal((1, 64)).astype(np.float32)
        csf = BatchedCSR.from_coo_list([a[0] for a in adjf], rows=32, cols=32, device=dev)
        txf, twf, tbf = t32(xf).requires_grad_(True), t32(wf).requires_grad_(True), t32(bf).requires_grad_(True)
        of = ops.graphconv_fused(txf, twf, tbf, csf)
        of.backward(t32(gf))
        ctx = ("pairs backward", Tf)
        check("full fwd", of, K.graphconv_fwd_fast(xf, adjf, [wf], [bf]), ctx=ctx)
        dxf, dwf, dbf = K.graphconv_bwd_fast(xf, adjf, [wf], gf)
        check("full dx", txf.grad, dxf, ctx=ctx); check("full dw", twf.grad, dwf[0], rel=1e-5, atol=1e-5 * float(np.abs(dwf[0]).max()), ctx=ctx)
        check("full db", tbf.grad, dbf[0], rel=1e-5, atol=1e-5 * float(np.abs(dbf[0]).max()), ctx=ctx)
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
    # ---- wide dense layer read out by GraphGather at 20,000 .. 120,000 node rows (several tiles per workgroup of the table GEMM) ----
    if case % 4 == 0:
        Ng = int(rng.choice([4, 7, 10, 32, 50]))
        Tg = int(rng.integers(20_000, 120_000)) // Ng
        dig, dog = [(256, 256), (84, 256), (300, 256), (256, 512), (128, 256), (256, 132)][int(rng.integers(0, 6))]
        actg = ["sigmoid", "relu", "tanh"][int(rng.integers(0, 3))]
        tee = bool(rng.integers(0, 2))
        xg = rng.standard_normal((Tg, Ng, dig)).astype(np.float32)
        wg = (rng.standard_normal((dig, dog)) / np.sqrt(dig)).astype(np.float32)
        bg = rng.standard_normal(dog).astype(np.float32)
        gpg, gyg = rng.standard_normal((Tg, dog)).astype(np.float32), rng.standard_normal((Tg, Ng, dog)).astype(np.float32)
        txg, twg, tbg = (t32(a_).requires_grad_(True) for a_ in (xg, wg, bg))
        yg, pg = ops.dense_gather(txg, twg, tbg, activation=actg)
        ((pg * t32(gpg)).sum() + ((yg * t32(gyg)).sum() if tee else 0.0)).backward()
        x64g = xg.astype(np.float64).reshape(Tg * Ng, dig)
        fg = {"relu": lambda v: np.maximum(v, 0), "sigmoid": lambda v: 1 / (1 + np.exp(-v)), "tanh": np.tanh}[actg]
        yrg = fg(x64g @ wg.astype(np.float64) + bg)
        ygn = yg.detach().cpu().numpy().reshape(Tg * Ng, dog)
        dfg = {"relu": (ygn > 0) * 1.0, "sigmoid": yrg * (1 - yrg), "tanh": 1 - yrg * yrg}[actg]
        dpg = (np.repeat(gpg.astype(np.float64), Ng, axis=0) + (gyg.reshape(Tg * Ng, dog) if tee else 0.0)) * dfg
        ctx = ("dense_gather", Tg, Ng, dig, dog, actg, tee)
        check("dense_gather y", yg, yrg, atol=2e-6 * max(1.0, float(np.abs(yrg).max())), rel=2e-6, ctx=ctx)
        check("dense_gather pooled", pg, yrg.reshape(Tg, Ng, dog).sum(1), ctx=ctx)
        check("dense_gather dx", txg.grad, dpg @ wg.astype(np.float64).T, ctx=ctx)
        check("dense_gather dw", twg.grad, x64g.T @ dpg, ctx=ctx); check("dense_gather db", tbg.grad, dpg.sum(0), ctx=ctx)
    # ---- register-split weight gradients (wgradx.hip / wgradn.hip: m >= 4096, narrow input or narrow output), ragged last step --
    if case % 4 == 1:
        if rng.integers(0, 2):
            dir_, dor_ = int(rng.integers(65, 97)), int(rng.integers(129, 257))
        else:
            dir_, dor_ = 4 * int(rng.integers(32, 65)), int(rng.integers(1, 65))
        Mr = int(rng.integers(4096, 60000))
        actr = [None, "sigmoid", "relu", "tanh"][int(rng.integers(0, 4))]
        need_x = bool(rng.integers(0, 2))
        xr_ = rng.standard_normal((Mr, dir_)).astype(np.float32)
        wr_ = (rng.standard_normal((dir_, dor_)) / np.sqrt(dir_)).astype(np.float32)
        br_ = rng.standard_normal(dor_).astype(np.float32)
        gr_ = rng.standard_normal((Mr, dor_)).astype(np.float32)
        txr = t32(xr_).requires_grad_(need_x)
        twr, tbr = t32(wr_).requires_grad_(True), t32(br_).requires_grad_(True)
        yr_ = ops.dense(txr, twr, tbr, activation=actr)
        yr_.backward(t32(gr_))
        pre64 = xr_.astype(np.float64) @ wr_.astype(np.float64) + br_
        f_ = {None: lambda z: z, "sigmoid": lambda z: 1 / (1 + np.exp(-z)), "relu": lambda z: np.maximum(z, 0), "tanh": np.tanh}[actr]
        y64 = f_(pre64)
        yg_ = yr_.detach().cpu().numpy()
        d64 = {None: np.ones_like(y64), "sigmoid": y64 * (1 - y64), "relu": (yg_ > 0) * 1.0, "tanh": 1 - y64 ** 2}[actr]
        dp64 = gr_.astype(np.float64) * d64
        ctx = ("wgrad narrow", Mr, dir_, dor_, actr, need_x)
        check("narrow dense fwd", yr_, y64, ctx=ctx)
        check("narrow dense dw", twr.grad, xr_.astype(np.float64).T @ dp64, ctx=ctx)
        check("narrow dense db", tbr.grad, dp64.sum(0), ctx=ctx)
        if need_x:
            check("narrow dense dx", txr.grad, dp64 @ wr_.astype(np.float64).T, ctx=ctx)
    # ---- device-side batch assembly ----------------------------------------------------------------------------------
    sel = rng.integers(-1, T, size=int(rng.integers(0, 3 * T + 2)))
    got = csr.gather(sel)
    host = BatchedCSR.from_coo_list([adjs[s][0] if s >= 0 else (np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [N, N])
                                     for s in sel], rows=N, cols=N, device=dev) if len(sel) else None
    if host is not None and not (torch.equal(got.rowptr, host.rowptr) and torch.equal(got.cv, host.cv)):
        fails.append(("gather", (N, T, len(sel)), 0, 0))
    # ---- wide / odd-width aggregation on larger graphs: tile kernel in column slices, 4-wave tiles, float2 lanes, gather --
    N2, T2 = int(rng.integers(1, 65)), int(rng.integers(1, 40))
    D2 = int(rng.choice([2, 6, 50, 64, 96, 128, 130, 200, 256, 300, 512, int(rng.integers(1, 300))]))
    C2 = int(rng.integers(1, 4))
    chans = [rand_graphs(T2, N2, rng.uniform(0.03, 0.3), empty_every=int(rng.integers(0, 5))) for _ in range(C2)]
    csrs = [BatchedCSR.from_coo_list([a[0] for a in ch], rows=N2, cols=N2, device=dev) for ch in chans]
    adj2 = BatchedAdjacency(csrs)
    rhs2 = rng.standard_normal((T2 * N2, C2 * D2)).astype(np.float32)
    g2 = rng.standard_normal((T2 * N2, D2)).astype(np.float32)
    actn = [None, "sigmoid", "relu", "tanh"][int(rng.integers(0, 4))]
    tr = t32(rhs2).requires_grad_(True)
    o2 = ops.bconv(adj2, tr, D2, activation=actn)
    o2.backward(t32(g2))
    pre = np.zeros((T2, N2, D2))
    r3 = rhs2.astype(np.float64).reshape(T2, N2, C2 * D2)
    for c in range(C2):
        pre += np.stack(K.bspmm([a[0] for a in chans[c]], list(r3[:, :, c * D2:(c + 1) * D2])))
    f_act = {None: lambda z: z, "sigmoid": lambda z: 1 / (1 + np.exp(-z)), "relu": lambda z: np.maximum(z, 0), "tanh": np.tanh}[actn]
    yref = f_act(pre)
    # relu': the mask of the kernel's OWN output (as tf.nn.relu's gradient reads its output): a pre-activation within one
    # rounding of 0 may legitimately land on either side
    d_act = {None: np.ones_like(yref), "sigmoid": yref * (1 - yref),
             "relu": (o2.detach().cpu().numpy().reshape(yref.shape) > 0) * 1.0, "tanh": 1 - yref ** 2}[actn]
    ctx = ("bconv+act", N2, D2, T2, C2, actn)
    check("bconv act fwd", o2, yref.reshape(T2 * N2, D2), ctx=ctx)
    gpre = (g2.reshape(T2, N2, D2) * d_act)
    drhs = np.concatenate([np.stack(K.bspmm([a[0] for a in chans[c]], list(gpre), adjoint_a=True)) for c in range(C2)], axis=2)
    check("bconv act bwd", tr.grad, drhs.reshape(T2 * N2, C2 * D2), ctx=ctx)
    # ---- activated dense layers incl. the wide-layer table / fused d-activation routes (m >= 1024) -------------------------
    M3 = int(rng.integers(1, 6000))
    di3, do3 = int(rng.choice([50, 81, 192, 256, 320, int(rng.integers(1, 400))])), int(rng.choice([12, 50, 129, 200, 256, 512, int(rng.integers(1, 400))]))
    act3 = ["sigmoid", "relu", "tanh"][int(rng.integers(0, 3))]
    x3 = rng.standard_normal((M3, di3)).astype(np.float32)
    w3 = K.glorot_uniform(rng, di3, do3); b3 = rng.standard_normal(do3).astype(np.float32)
    g3 = rng.standard_normal((M3, do3)).astype(np.float32)
    tx3, tw3, tb3 = t32(x3).requires_grad_(True), t32(w3).requires_grad_(True), t32(b3).requires_grad_(True)
    y3 = ops.dense(tx3, tw3, tb3, activation=act3)
    y3.backward(t32(g3))
    z3 = x3.astype(np.float64) @ w3.astype(np.float64) + b3
    a3 = {"sigmoid": 1 / (1 + np.exp(-z3)), "relu": np.maximum(z3, 0), "tanh": np.tanh(z3)}[act3]
    gz3 = g3 * {"sigmoid": a3 * (1 - a3), "relu": (y3.detach().cpu().numpy() > 0) * 1.0, "tanh": 1 - a3 ** 2}[act3]
    ctx = ("dense+act", M3, di3, do3, act3)
    check("dense act fwd", y3, a3, ctx=ctx); check("dense act dx", tx3.grad, gz3 @ w3.astype(np.float64).T, ctx=ctx)
    check("dense act dw", tw3.grad, x3.astype(np.float64).T @ gz3, rel=5e-5, ctx=ctx); check("dense act db", tb3.grad, gz3.sum(0), rel=5e-5, ctx=ctx)
    # ---- round 3: ragged-compact layer chain vs the oracle's PADDED formulation ---------------------------------------------
    from kgcn_amd import ragged as RG
    Br, Nr = int(rng.integers(1, 60)), int(rng.integers(2, 40))
    Fr, W1, W2 = int(rng.integers(1, 90)), int(rng.choice([7, 50, 64, 130, 256])), int(rng.choice([5, 50, 64]))
    szr = rng.integers(0, Nr + 1, size=Br)
    adjr, xr = [], np.zeros((Br, Nr, Fr), np.float32)
    for b_, n_ in enumerate(szr):
        if n_ == 0:
            adjr.append([(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [Nr, Nr])]); continue
        a_ = (rng.random((n_, n_)) < 0.3) * rng.standard_normal((n_, n_))
        ii = np.argwhere(a_ != 0).astype(np.int32)
        adjr.append([(ii, a_[a_ != 0].astype(np.float32), [Nr, Nr])])
        xr[b_, :n_] = rng.standard_normal((n_, Fr))
    actr = [None, "sigmoid", "relu", "tanh"][int(rng.integers(0, 4))]
    layers.aggregate_first = bool(rng.integers(0, 2))
    c1, d1 = layers.GraphConv(W1, 1, activation=actr), layers.GraphDense(W2, activation="sigmoid")
    rb = RG.compact(t32(xr), adjr, szr)
    h_r = d1(c1(rb.features, adj=rb))
    pooled_r = layers.GraphGather()(h_r, ragged=rb)
    w1n, b1n = rng.standard_normal((1, W1)).astype(np.float32) * 0.2, None
    with torch.no_grad():
        c1.bias[0].copy_(t32(w1n)); d1.bias.copy_(t32(rng.standard_normal(W2) * 0.2))
    txr = t32(xr).requires_grad_(True)
    rb = RG.compact(txr, adjr, szr)
    pooled_r = layers.GraphGather()(d1(c1(rb.features, adj=rb)), ragged=rb)
    gr = rng.standard_normal((Br, W2)).astype(np.float32)
    pooled_r.backward(t32(gr))
    p_ = lambda t: t.detach().cpu().numpy()
    f_a = {None: lambda z: z, "sigmoid": lambda z: 1 / (1 + np.exp(-z)), "relu": lambda z: np.maximum(z, 0), "tanh": np.tanh}[actr]
    pre1 = K.graphconv_fwd_fast(xr, adjr, [p_(c1.w[0])], [p_(c1.bias[0])])
    a1 = f_a(pre1)
    z2 = K.graphdense_fwd(a1, p_(d1.kernel), p_(d1.bias)); a2 = 1 / (1 + np.exp(-z2))
    ctx = ("ragged chain", Br, Nr, Fr, W1, W2, actr, layers.aggregate_first)
    check("ragged pooled", pooled_r, K.gather_fwd(a2), ctx=ctx)
    if actr != "relu":                       # relu masks of pre-activations near 0 may differ between fp32 and fp64
        dz2 = K.gather_bwd(gr, Nr) * a2 * (1 - a2)
        da1, dk2, dc2 = K.graphdense_bwd(a1, p_(d1.kernel), dz2)
        dpre1 = da1 * {None: 1.0, "sigmoid": a1 * (1 - a1), "tanh": 1 - a1 ** 2}[actr]
        dxr, dw1, db1 = K.graphconv_bwd_fast(xr, adjr, [p_(c1.w[0])], dpre1)
        check("ragged dK2", d1.kernel.grad, dk2, rel=5e-5, ctx=ctx); check("ragged dc2", d1.bias.grad, dc2, rel=5e-5, ctx=ctx)
        check("ragged dW1", c1.w[0].grad, dw1[0], rel=5e-5, ctx=ctx); check("ragged db1", c1.bias[0].grad, db1[0], rel=5e-5, ctx=ctx)
        check("ragged dx", txr.grad, dxr, rel=5e-5, ctx=ctx)
    layers.aggregate_first = True
    # ---- round 3: GINAggregate with epsilon (d eps rides in the adjoint aggregation) -------------------------------------------
    Dg = int(rng.choice([4, 8, 50, 64, 128, 256, int(rng.integers(1, 130))]))
    xg = rng.standard_normal((T, N, Dg)).astype(np.float32)
    gg = rng.standard_normal((T, N, Dg)).astype(np.float32)
    gin = layers.GINAggregate(1)
    txg2 = t32(xg).requires_grad_(True)
    gin(txg2, adj=csr)
    with torch.no_grad():
        gin.epsilon[0].fill_(0.3)
    og2 = gin(txg2, adj=csr)
    og2.backward(t32(gg))
    check("gin fwd", og2, K.gin_fwd(xg, adjs, [0.3]), ctx=("gin", N, Dg, T))
    dxg2, deps2 = K.gin_bwd(xg, adjs, [0.3], gg)
    check("gin dx", txg2.grad, dxg2, ctx=("gin", N, Dg, T))
    sc_ = float(np.abs(gg.astype(np.float64) * xg).sum()) + 1e-30
    if abs(float(gin.epsilon[0].grad) - float(deps2[0])) > 3e-6 * sc_:
        fails.append(("gin deps", (N, Dg, T), float(gin.epsilon[0].grad), float(deps2[0])))
    # ---- round 3: cross-layer stack (model.py network) vs the same modules layer by layer -------------------------------------
    from kgcn_amd import models as MD
    Bs, Ns = int(rng.integers(1, 50)) if case % 4 else int(rng.integers(300, 900)), int(rng.integers(1, 33))
    Fs, Ws = int(rng.integers(1, 65)), int(rng.integers(1, 57))
    szs = rng.integers(0, Ns + 1, size=Bs)
    dup_s = case % 3 == 0                            # duplicate entries: rows longer than N take the tile kernels' CSR walk
    ops.stack_route = (0, 1, 2)[case % 3]
    adjs_s, xs = [], np.zeros((Bs, Ns, Fs), np.float32)
    for b_, n_ in enumerate(szs):
        if n_ == 0:
            adjs_s.append([(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [Ns, Ns])]); continue
        a_ = (rng.random((n_, n_)) < 0.3) * rng.standard_normal((n_, n_))
        ix_, vl_ = np.argwhere(a_ != 0).astype(np.int32), a_[a_ != 0].astype(np.float32)
        if dup_s and len(ix_):
            k_ = rng.integers(0, len(ix_), size=2 * len(ix_))
            ix_, vl_ = np.concatenate([ix_, ix_[k_]]), np.concatenate([vl_, vl_[k_]])
        adjs_s.append([(ix_, vl_, [Ns, Ns])])
        xs[b_, :n_] = rng.standard_normal((n_, Fs))
    use_en = bool(rng.integers(0, 2))
    outs_s = {}
    for fused_s in (True, False, "route1"):
        if fused_s == "route1":
            if not os.environ.get("FUZZ_DEBUG"):
                continue
            ops.stack_route = 1
        layers.stack_fusion = bool(fused_s)
        torch.manual_seed(case)
        md = MD.GCN(1, 2).to(dev)
        for m_ in (md.conv1, md.conv2, md.conv3, md.dense):
            m_.output_dim = Ws
        txs = t32(xs).requires_grad_(True)
        en_s = torch.as_tensor(szs) if use_en else None
        md(txs, adjs_s, enabled_node_nums=en_s)
        with torch.no_grad():
            gen_ = torch.Generator(device="cpu").manual_seed(case)
            for p__ in md.parameters():
                if p__.dim() == 1 or p__.shape[0] == 1:
                    p__.copy_(torch.randn(p__.shape, generator=gen_).to(dev) * 0.2 + (1.0 if p__ is md.bn.gamma else 0.0))
            md.bn.moving_mean.copy_(torch.randn(Ws, generator=gen_).to(dev) * 0.1)
            md.bn.moving_variance.copy_(torch.rand(Ws, generator=gen_).to(dev) + 0.5)
        lg = md(txs, adjs_s, enabled_node_nums=en_s)
        lg.sum().backward()
        outs_s[fused_s] = [lg.detach(), txs.grad] + [p__.grad for p__ in md.parameters()]
    layers.stack_fusion = True
    ops.stack_route = 0
    nf_ = len(fails)
    for i_, (a_, b_) in enumerate(zip(outs_s[True], outs_s[False])):
        # duplicate entries triple some adjacency values: saturated sigmoids make y (1 - y) ill-conditioned in fp32 and the three
        # implementations (layers, one-graph stack, tile stack) then differ by ~1e-4 from EACH OTHER on such a graph (FUZZ_DEBUG=1)
        check("stack vs layers #%d" % i_, a_, b_.cpu().numpy(), rel=1e-3 if dup_s else 5e-5, atol=5e-6,
              ctx=("stack", Bs, Ns, Fs, Ws, use_en, dup_s, case % 3))
    if len(fails) > nf_ and os.environ.get("FUZZ_DEBUG"):
        d_ = (outs_s[True][1] - outs_s[False][1]).abs()
        bi_ = int(d_.reshape(Bs, -1).max(dim=1).values.argmax())
        print("DEBUG stack case", case, "worst graph", bi_, "size", int(szs[bi_]), "nnz", len(adjs_s[bi_][0][1]),
              "max|dfeat| of that graph", float(outs_s[False][1][bi_].abs().max()), "err", float(d_[bi_].max()),
              "route 1 vs layers on that graph:", float((outs_s["route1"][1] - outs_s[False][1]).abs()[bi_].max()),
              "route 1 vs tiles:", float((outs_s["route1"][1] - outs_s[True][1]).abs()[bi_].max()),
              "graphs with err > 1e-5:", int((d_.reshape(Bs, -1).max(dim=1).values > 1e-5).sum()),
              "max |adj val|", max((float(np.abs(a__[0][1]).max()) if len(a__[0][1]) else 0.0) for a__ in adjs_s))
    # ---- mini-batch assembly in two launches vs the per-container gather ----------------------------------------------------------
    from kgcn_amd import data_util as DU
    Ga, Na, Ta = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 700))
    mats_a = [a[0] for a in rand_graphs(Ga, Na, rng.uniform(0.0, 0.5), empty_every=int(rng.integers(0, 5)), dup=bool(rng.integers(0, 2)))]
    fa = rng.standard_normal((Ga, Na, 3)).astype(np.float32)
    dsa = DU.DeviceGraphDataset([DU.FlatAdjacency.from_coo_list(mats_a, n_nodes=Na)], fa, device=dev)
    sba = dsa.static_batch(Ta)
    taba = t32(rng.standard_normal((Ga, 5)).astype(np.float32))
    tab_s = sba.add_table(taba)
    nsel = int(rng.integers(0, Ta + 1))
    ia = rng.integers(0, Ga, nsel)
    sba.load(ia)
    sela = np.full(Ta, -1, np.int64); sela[:nsel] = ia
    for pairs_ in sba._sources:
        for src_, st_ in pairs_:
            ref_ = src_.gather(sela)
            n_ = int(ref_.rowptr[-1])
            if not (torch.equal(st_.rowptr, ref_.rowptr) and torch.equal(st_.cv[:n_], ref_.cv[:n_]) and
                    (not src_.row_pad or torch.equal(st_.slots, ref_.slots))):
                fails.append(("batch_assemble container", (Ga, Na, Ta, nsel, src_.row_pad), 0.0, 0.0))
    sd_ = torch.from_numpy(np.maximum(sela, 0)).to(dev)
    vd_ = torch.from_numpy(sela >= 0).to(dev)
    if not (torch.equal(sba.features, dsa.features[sd_] * vd_[:, None, None]) and torch.equal(tab_s, taba[sd_] * vd_[:, None])):
        fails.append(("batch_assemble tables", (Ga, Na, Ta, nsel), 0.0, 0.0))
    # ---- device-side COO pack -----------------------------------------------------------------------------------------------
    Tp, Np, nz = int(rng.integers(1, 60)), int(rng.integers(1, 65)), int(rng.integers(0, 4000))
    gp, rp_, cp = rng.integers(0, Tp, nz), rng.integers(0, Np, nz), rng.integers(0, Np, nz)
    vp = rng.standard_normal(nz).astype(np.float32)
    hostp = BatchedCSR.from_arrays(gp, rp_, cp, vp, Tp, Np, Np, device=dev)
    ti = lambda a: torch.from_numpy(np.asarray(a, np.int32)).to(dev)
    devp = BatchedCSR.from_device_coo(ti(gp), ti(rp_), ti(cp), t32(vp), Tp, Np, Np)
    pairs = [(devp, hostp), (devp.transpose(), hostp.transpose())]
    if Np <= 32:
        try:
            hp4 = hostp.padded4()
        except ValueError:
            hp4 = None                                     # a row longer than 252 padded entries: both packers refuse
        try:
            dp4 = devp.padded4()
        except ValueError:
            dp4 = None
        if (hp4 is None) != (dp4 is None):
            fails.append(("coo pack: only one packer refused", (Tp, Np, nz), 0, 0))
        elif hp4 is not None:
            pairs += [(dp4, hp4)]
    for a_, b_ in pairs:
        same = torch.equal(a_.rowptr, b_.rowptr) and torch.equal(a_.cv, b_.cv) and a_.max_nnz == b_.max_nnz
        if a_.row_pad:
            same = same and torch.equal(a_.slots, b_.slots) and torch.equal(a_.graph_ptr, b_.graph_ptr)
        if not same:
            fails.append(("coo pack", (Tp, Np, nz, a_.row_pad), 0, 0))
print("%d cases in %.1f s, %d failures" % (CASES, time.time() - t_start, len(fails)))
for f in fails[:30]:
    print("  FAIL", f)
sys.exit(1 if fails else 0)
