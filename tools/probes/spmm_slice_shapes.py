"""ops.bspmm at several (graphs, nodes, width) shapes: time per launch; run with KGCN_SPMM_SLICES=1 / 2 / 4 on a DEV_KNOBS build"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from kgcn_amd import ops, BatchedCSR
dev = torch.device("cuda:0")

def timed(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

res = {}
for T, N in ((100_000, 32), (40_000, 50), (200_000, 10), (100_000, 16)):
    if N == 32:
        g, r, c = bench.gen_mol_graphs(T, seed=1)
    elif N == 10:
        g, r, c, _, _ = bench.gen_ring_graphs(T, N, seed=1)
    else:
        _, g, r, c, _ = bench.gen_tox21_like(T, N, seed=1)
    csr = BatchedCSR.from_arrays(g, r, c, np.ones(g.shape[0], np.float32), T, N, N, device=dev)
    for d in (32, 64, 128, 256):
        if T * N * d * 4 > 3e9:
            continue
        x = torch.randn(T, N, d, device=dev)
        us = timed(lambda: ops.bspmm(csr, x))
        byt = 8 * T * N * d + 8 * g.shape[0] + 4 * T * (N + 1)
        res["T=%d N=%d d=%d" % (T, N, d)] = [round(us, 1), round(byt / us / 1e3 / 8000, 3)]
print(json.dumps(res))
