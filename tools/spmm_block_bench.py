#!/usr/bin/env python3
"""Aggregation over a cfg4-shaped ragged-compact batch (4,096 Tox21-like molecules of 5..50 atoms, ~113,000 rows) through the C
ABI: forward (kgcn_bconv_act_f32) and adjoint with the activation derivative (kgcn_bspmm_dact_f32), block kernel vs row-chunk
kernel (the same container without its block table), operands rotated through four buffer sets (> 256 MB, the Infinity Cache).
usage: python tools/spmm_block_bench.py [--d 256,50,84]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from kgcn_amd import data_util as D                              # noqa: E402
from kgcn_amd._lib import lib, ptr, current_stream, check        # noqa: E402
from kgcn_amd.batched_csr import BatchedCSR                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", default="256,50,84")
ap.add_argument("--reps", type=int, default=24)
args = ap.parse_args()
dev = torch.device("cuda:0")
G, N, B = 8192, 50, 4096
sizes, g, r, c, rng = bench.gen_tox21_like(G, N, seed=4)
chan = D.normalize_adj(D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), G, N))
ds = D.DeviceGraphDataset([chan], None, device=dev, sizes=sizes)
sb = ds.static_ragged_batch(B)
sb.load(np.arange(B))
rb = sb.ragged
cap = rb.capacity
a = rb.adjacency.channels[0]
at = a.transpose()
plain = lambda x: BatchedCSR(x.rowptr, x.cv, 1, cap, cap, x.max_nnz)
R = int(rb.row_count.item())
nnz = int(a.rowptr[-1].item())
out = {"rows": R, "capacity": cap, "nnz": nnz}


def timeit(fn, nsets):
    for i in range(4):
        fn(i % nsets)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for i, (s, e) in enumerate(ev):
        s.record(); fn(i % nsets); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return 1e3 * ts[len(ts) // 2]


for d in [int(v) for v in args.d.split(",")]:
    nsets = 4
    xs = [torch.randn((cap, d), device=dev) for _ in range(nsets)]
    ys = [torch.sigmoid(torch.randn((cap, d), device=dev)) for _ in range(nsets)]
    os_ = [torch.empty((cap, d), device=dev) for _ in range(nsets)]
    res = {}
    for name, cf, ct in (("block", a, at), ("rows", plain(a), plain(at))):
        f = lambda i: check(lib.kgcn_bconv_act_f32(cf.desc(), 1, ptr(xs[i]), d, cap * d, 0, d, ptr(os_[i]), d, cap * d, 1,
                                                   current_stream()))
        b = lambda i: check(lib.kgcn_bspmm_dact_f32(ct.desc(), ptr(xs[i]), ptr(ys[i]), d, cap * d, d, 1, ptr(os_[i]), d, cap * d,
                                                    0.0, current_stream()))
        res[name + "_fwd_us"] = round(timeit(f, nsets), 1)
        res[name + "_adj_us"] = round(timeit(b, nsets), 1)
    alg_f = 2 * cap * d * 4 + 8 * nnz + 4 * cap
    res["fwd_alg_MB"] = round(alg_f / 1e6, 1)
    res["block_fwd_frac"] = round(alg_f / res["block_fwd_us"] / 8e6, 3)
    res["block_adj_frac"] = round((alg_f + cap * d * 4) / res["block_adj_us"] / 8e6, 3)
    out["d%d" % d] = res
    print(d, res, file=sys.stderr)
print(json.dumps(out))
