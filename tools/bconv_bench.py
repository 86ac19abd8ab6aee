#!/usr/bin/env python3
"""Multi-channel GraphConv / Bconv at HEAD (north_star names bconv_call): the split_adj_flag case of the
reference -- one adjacency channel per bond type, kgcn/data_util.py:76-122, op contract kgcn/bconv_call.py:11-23 -- with C = 6
channels on (a) a synthetic.jbl-shaped batch (N = 10, D = 50) and (b) a cfg2-shaped batch (N = 32, D = 64).  One JSON line per
shape: the layer's forward + backward (GEMM -> FW [B N, C Dout] -> kgcn_bconv_act_f32; backward: adjoint Bconv, dX / dW GEMMs)
and the Bconv launches alone, priced against 8 TB/s on their ALGORITHMIC bytes
    kgcn_bconv_f32 (forward)   FW in: C * 4 N D   + the C CSR slices + out: 4 N D          per graph
    adjoint (backward)         g in: 4 N D        + the C CSR slices + dFW out: C * 4 N D  per graph
    a channel-fused layer would move (what `fused_layer_bytes` reports): x in 4 N Din + C CSRs + out 4 N Dout per direction.
The edges of the cfg2 generator (tree + 3 extra edges) are dealt to the channels at random (a bond type per edge, symmetric);
the self loops form channel 0 together with its share -- every channel's adjacency is symmetric, as build_adjs' are.
usage: python tools/bconv_bench.py [--graphs T] [--channels C] [--shape jbl|cfg2|both] [--steps K] [--profile]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kgcn_amd import BatchedCSR, layers, ops  # noqa: E402
from kgcn_amd.batched_csr import BatchedAdjacency  # noqa: E402

HBM = 8000.0


def split_channels(g, r, c, T, n, C, rng):
    """bond type per undirected edge; self loops -> channel 0"""
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    key = (g * n + lo) * n + hi
    uniq, inv = np.unique(key, return_inverse=True)
    ch_of = rng.integers(0, C, size=uniq.shape[0])
    ch = ch_of[inv]
    ch[r == c] = 0
    return ch


def run(shape, T, C, steps, profile):
    dev = torch.device("cuda:0")
    n, d = (10, 50) if shape == "jbl" else (32, 64)
    rng = np.random.default_rng(77)
    g, r, c = bench.gen_mol_graphs(T, n=n, extra=2 if n == 10 else 3, seed=99)
    ch = split_channels(g, r, c, T, n, C, rng)
    chans, csr_bytes = [], 0.0
    for k in range(C):
        m = ch == k
        chans.append(BatchedCSR.from_arrays(g[m], r[m], c[m], np.ones(int(m.sum()), np.float32), T, n, n, device=dev))
        csr_bytes += 4.0 * (n + 1) + 8.0 * m.sum() / T
    adj = BatchedAdjacency(chans)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    x = torch.randn((T, n, d), device=dev, generator=gen).requires_grad_(True)
    gy = torch.randn((T, n, d), device=dev, generator=gen)
    layer = layers.GraphConv(d, C, activation="sigmoid").to(dev)      # (the models' tf.sigmoid(layer) rides in the aggregation's epilogue)
    layer.build((T, n, d), dev)

    def step():
        x.grad = None
        for p in layer.parameters():
            p.grad = None
        layer(x, adj=adj).backward(gy)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    res = {"shape": shape, "graphs": T, "n_nodes": n, "d": d, "channels": C, "nnz_per_graph_all_channels": float(g.shape[0]) / T,
           "layer_fwd_bwd_ms": {"median": ms[len(ms) // 2], "p10": ms[len(ms) // 10], "p90": ms[(9 * len(ms)) // 10]},
           "graphs_per_s": T / (ms[len(ms) // 2] * 1e-3)}
    if profile:
        return res
    # the Bconv launches alone: forward out = sum_c A_c FW_c (FW [T n, C d]) and the adjoint dFW_c = A_c^T g
    fw = torch.randn((T * n, C * d), device=dev, generator=gen)
    out = torch.empty((T * n, d), device=dev)
    dfw = torch.empty_like(fw)
    g2 = gy.reshape(T * n, d)
    from kgcn_amd._lib import lib, ptr, current_stream, check
    adj_t = BatchedAdjacency([c_.transpose() for c_ in chans])

    def fwd():
        check(lib.kgcn_bconv_act_f32(adj.desc_array(False), C, ptr(fw), C * d, n * C * d, d, d, ptr(out), d, n * d, 0, current_stream()),
              "kgcn_bconv_act_f32")

    def adjoint():                                             # ONE launch: g read once, every channel's column block written (what _BConv.backward does)
        check(lib.kgcn_bconv_fanout_f32(adj.desc_array(True), C, ptr(g2), None, d, n * d, d, 0, ptr(dfw), C * d, n * C * d, d, current_stream()),
              "kgcn_bconv_fanout_f32")

    def adjoint_per_channel():                                 # rounds 2-5: one launch per channel, each reading g again
        for k in range(C):
            ops.bspmm_raw(adj_t.channels[k], g2, d, dfw, out_ld=C * d, out_gs=n * C * d, out_col=k * d)

    tm = {}
    for name, fn in (("bconv_forward", fwd), ("bconv_adjoint", adjoint), ("bconv_adjoint_per_channel_launches", adjoint_per_channel)):
        for _ in range(5):
            fn()
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in e:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in e)
        tm[name] = t[len(t) // 2]
    per_dir = 4.0 * n * d * (C + 1) + csr_bytes
    res["algorithmic_bytes_per_graph"] = {"bconv_forward": per_dir, "bconv_adjoint": per_dir, "csr_all_channels": csr_bytes,
                                          "note": "kernels: bconv_loop_kernel (forward), bconv_fanout_kernel (adjoint); rounds 2-5: spmm_tile_kernel with all "
                                                  "channels staged at once / one launch per channel",
                                          "fused_layer_bytes_per_direction": 8.0 * n * d + csr_bytes}
    for k, v in tm.items():
        gbs = per_dir * T / (v * 1e-3) / 1e9
        res[k] = {"median_ms": v, "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=0)
    ap.add_argument("--channels", type=int, default=6)
    ap.add_argument("--shape", default="both")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--profile", action="store_true", help="layer steps only (for rocprofv3 passes)")
    a = ap.parse_args()
    for shape in (("jbl", "cfg2") if a.shape == "both" else (a.shape,)):
        T = a.graphs or (200_000 if shape == "jbl" else 50_000)
        print(json.dumps(run(shape, T, a.channels, a.steps, a.profile)), flush=True)


if __name__ == "__main__":
    main()
