#!/usr/bin/env python3
"""Per-kernel roofline measurements on the cfg2 workload (GPU box): HIP-event timed launches of
every kernel of the path, algorithmic bytes / flops per launch, fraction of the 8 TB/s HBM peak
(or of the 157.3 TF fp32 MFMA peak for the dense contraction).  Prints one JSON object.
usage: python tools/bench_kernels.py [graphs] [reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_cfg2, algorithmic_bytes, HBM_PEAK_GBS  # noqa: E402
from kgcn_amd import ops  # noqa: E402
from kgcn_amd._lib import lib, ptr, current_stream, check  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
wl = make_cfg2(T, dev)
csr, x, g, w, b = wl["csr"], wl["x"], wl["g"], wl["w"], wl["bias"].reshape(-1)
N, D, NNZ = 32, 64, wl["nnz_per_graph"]
x2d, g2d = x.reshape(T * N, D), g.reshape(T * N, D)
out = torch.empty_like(x2d)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
db = torch.empty(D, device=dev)
ab = algorithmic_bytes(N, D, D, NNZ)
MFMA_PEAK_TF = 157.3


def timeit(fn, reps=REPS, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b_ in ev:
        a.record()
        fn()
        b_.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) for a, b_ in ev)
    return {"median_ms": ts[len(ts) // 2], "p10_ms": ts[len(ts) // 10], "p90_ms": ts[(9 * len(ts)) // 10]}


res = {"graphs": T, "n_nodes": N, "d": D, "nnz_per_graph": NNZ, "hbm_peak_GBs": HBM_PEAK_GBS}


def add(name, fn, bytes_per_graph=None, flops_per_graph=None):
    t = timeit(fn)
    r = dict(t)
    s = t["median_ms"] * 1e-3
    r["graphs_per_s"] = T / s
    if bytes_per_graph:
        r["alg_bytes_per_graph"] = bytes_per_graph
        r["GBs"] = bytes_per_graph * T / s / 1e9
        r["frac_hbm_peak"] = r["GBs"] / HBM_PEAK_GBS
    if flops_per_graph:
        r["TFs"] = flops_per_graph * T / s / 1e12
        r["frac_mfma_f32_peak"] = r["TFs"] / MFMA_PEAK_TF
    res[name] = r


spmm_bytes = 2 * 4 * N * D + ab["csr"]
add("bspmm_fwd (spmm_tile_kernel)", lambda: ops.bspmm_raw(csr, x2d, D, out), spmm_bytes, 2 * NNZ * D)
csr_t = csr.transpose()
add("bspmm_adjoint (spmm_tile_kernel, A^T)", lambda: ops.bspmm_raw(csr_t, g2d, D, out), spmm_bytes, 2 * NNZ * D)
add("dense_fwd (x@W+b)", lambda: check(lib.kgcn_dense_fwd_f32(ptr(x2d), T * N, D, D, ptr(w), D, 0, ptr(b), ptr(out), D, D,
                                                              current_stream())), 2 * 4 * N * D, 2 * N * D * D)
wsb = lib.kgcn_dense_wgrad_workspace_bytes(T * N, D, D)
wsp = torch.empty(wsb // 4, device=dev)
add("dense_wgrad (x^T@dy, colsum)", lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x2d), D, ptr(g2d), D, T * N, D, D, ptr(dw), ptr(db),
                                                                           ptr(wsp), wsb, current_stream())),
    2 * 4 * N * D, 2 * N * D * D)
p4, p4t = csr.padded4(), csr_t.padded4()
o3 = out.reshape(T, N, D)
add("graphconv_fwd fused", lambda: check(lib.kgcn_graphconv_fwd_f32(p4.desc(), ptr(x), ptr(w), ptr(b), D, D, ptr(o3),
                                                                    current_stream())), ab["fwd"], 2 * N * D * D + 2 * NNZ * D)
wsb2 = lib.kgcn_graphconv_bwd_workspace_bytes(T, D, D)
wsp2 = torch.empty(wsb2 // 4, device=dev)
add("graphconv_bwd fused (+1 reduce launch)",
    lambda: check(lib.kgcn_graphconv_bwd_f32(p4t.desc(), ptr(x), ptr(w), ptr(g), D, D, ptr(dx), ptr(dw), ptr(db),
                                             ptr(wsp2), wsb2, current_stream())), ab["bwd"], 4 * N * D * D + 2 * NNZ * D)
# device-side mini-batch assembly: T graphs gathered (shuffled) out of a resident dataset of T graphs;
# bytes = read + write of rowptr and cv
sel = np.random.default_rng(0).permutation(T)
csr.graph_counts()
gather_bytes = 2 * (4 * N + 8 * NNZ)
add("batch assembly A (kgcn_csr_gather_graphs)", lambda: csr.gather(sel), gather_bytes)
p4.graph_counts()
p4_bytes = 2 * (8 * N + 8 * (p4.nnz // T))
add("batch assembly A row-padded + slots", lambda: p4.gather(sel), p4_bytes)
# achievable HBM copy bandwidth on this box (float4 copy of 2 x 819 MB), the practical ceiling
src, dst = x.reshape(-1), torch.empty_like(x).reshape(-1)
t = timeit(lambda: dst.copy_(src))
res["device_copy"] = dict(t, GBs=2 * src.numel() * 4 / (t["median_ms"] * 1e-3) / 1e9)
print(json.dumps(res, indent=1))
