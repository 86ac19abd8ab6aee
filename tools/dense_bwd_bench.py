#!/usr/bin/env python
"""One-pass backward of a wide dense layer (kgcn_dense_bwd_f32, csrc/gemmb.hip) against the two-kernel route it replaces
(kgcn_dense_dx_dact_f32 / kgcn_dense_fwd_f32(trans) + kgcn_dense_wgrad_f32), through autograd (ops.dense), at the row counts
of BASELINE configs 4 and 5.  Each route: 30 backward passes between two HIP events after 5 warm-up passes.
usage: python tools/dense_bwd_bench.py [rows ...]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from kgcn_amd import ops  # noqa: E402


def run(m, act, fused, reps=30):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((m, 256), device=dev, generator=g).requires_grad_(True)
    w = (torch.randn((256, 256), device=dev, generator=g) * 0.06).requires_grad_(True)
    b = torch.zeros(256, device=dev).requires_grad_(True)
    gy = torch.randn((m, 256), device=dev, generator=g)
    ops.dense_bwd_fusion = fused
    ops.weight_tables.refresh()
    y = ops.dense(x, w, b, activation=act)
    ops.weight_tables.refresh()
    def bwd():
        x.grad = w.grad = b.grad = None
        y.backward(gy, retain_graph=True)
    for _ in range(5):
        bwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    ops.dense_bwd_fusion = True
    return e0.elapsed_time(e1) / reps * 1e3


def run_gather(m, fused, passed_on, reps=30):
    """GraphDense(256, relu) read out by GraphGather (model_gin.py:45-60): gradient = rows + pooled broadcast (or pooled alone)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(2)
    N = 10
    T = m // N
    x = torch.randn((T, N, 256), device=dev, generator=g).requires_grad_(True)
    w = (torch.randn((256, 256), device=dev, generator=g) * 0.06).requires_grad_(True)
    b = torch.zeros(256, device=dev).requires_grad_(True)
    gy = torch.randn((T, N, 256), device=dev, generator=g)
    gp = torch.randn((T, 256), device=dev, generator=g)
    ops.dense_bwd_fusion = fused
    ops.weight_tables.refresh()
    y, pooled = ops.dense_gather(x, w, b, activation="relu")
    ops.weight_tables.refresh()
    outs, grads = ([y, pooled], [gy, gp]) if passed_on else ([pooled], [gp])
    def bwd():
        x.grad = w.grad = b.grad = None
        torch.autograd.backward(outs, grads, retain_graph=True)
    for _ in range(5):
        bwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    ops.dense_bwd_fusion = True
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    rows = [int(a) for a in sys.argv[1:]] or [117888, 200000]
    out = []
    for m in rows:
        for act in (None, "relu", "sigmoid"):
            two, one = run(m, act, False), run(m, act, True)
            passes = 3 + (1 if act else 0)
            rec = {"rows": m, "act": act, "two_kernel_us": round(two, 1), "one_pass_us": round(one, 1),
                   "one_pass_algorithmic_MB": round(passes * m * 1024 / 1e6, 1),
                   "one_pass_frac_of_8TBs": round(passes * m * 1024 / (one * 1e-6) / 8e12, 3)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        for passed_on in (True, False):
            two, one = run_gather(m, False, passed_on), run_gather(m, True, passed_on)
            passes = 3 + (1 if passed_on else 0)
            rec = {"rows": m, "act": "relu + read-out" + (" + rows" if passed_on else " alone"), "two_kernel_us": round(two, 1),
                   "one_pass_us": round(one, 1), "one_pass_algorithmic_MB": round(passes * m * 1024 / 1e6, 1),
                   "one_pass_frac_of_8TBs": round(passes * m * 1024 / (one * 1e-6) / 8e12, 3)}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
