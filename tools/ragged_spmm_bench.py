#!/usr/bin/env python3
"""Aggregation kernels on a cfg4-shaped batch in both layouts: padded [4096 x 50 x d] (LDS tile kernel) and ragged-compact
block-diagonal [R x d] (row-chunk kernel): forward with sigmoid epilogue (kgcn_bconv_act_f32) and adjoint with the activation
derivative (kgcn_bspmm_dact_f32), HIP-event medians, algorithmic bytes (rhs read once + out written once + CSR [+ act out])."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kgcn_amd import BatchedAdjacency, BatchedCSR, _lib, ops, ragged  # noqa: E402

dev = torch.device("cuda:0")
B, N = 4096, 50
sizes, g, r, c, rng = bench.gen_tox21_like(B, N, seed=4)
val = bench.kipf_values(g, r, c, B, N)
csr = BatchedCSR.from_arrays(g, r, c, val, B, N, N, device=dev)
adj = BatchedAdjacency([csr])
rb = ragged.compact(None, adj, sizes)
R, cap = rb.rows, rb.capacity
res = {"graphs": B, "valid_rows": R, "capacity": cap, "nnz": int(csr.nnz)}


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3


for d in [int(a) for a in sys.argv[1:]] or [256, 50, 84, 64]:
    for name, a, rows in (("padded", adj, B * N), ("ragged", rb.adjacency, cap)):
        x = torch.randn(rows, d, device=dev)
        y = torch.empty_like(x)
        gr = torch.randn(rows, d, device=dev)
        ch = a.channels[0]
        cht = ch.transpose()
        T, M = ch.num_graphs, ch.rows
        fwd = lambda: _lib.check(_lib.lib.kgcn_bconv_act_f32(a.desc_array(False), 1, _lib.ptr(x), d, M * d, d, d, _lib.ptr(y), d,
                                                            M * d, 1, _lib.current_stream()))
        plain = lambda: ops.bspmm_raw(ch, x, d, y)
        dact = lambda: _lib.check(_lib.lib.kgcn_bspmm_dact_f32(cht.desc(), _lib.ptr(gr), _lib.ptr(y), d, M * d, d, 1,
                                                              _lib.ptr(x), d, M * d, 0.0, _lib.current_stream()))
        csr_b = 8 * int(csr.nnz) + 4 * rows
        for tag, fn, nb in (("fwd_sigmoid", fwd, 2), ("plain", plain, 2), ("adjoint_dact", dact, 3)):
            us = timed(fn)
            by = 4 * rows * d * nb + csr_b
            res["%s_d%d_%s" % (name, d, tag)] = {"us": round(us, 1), "GB_per_s": round(by / us / 1e3), "frac_hbm": round(by / us / 8e6, 3)}
print(json.dumps(res, indent=1))
