#!/usr/bin/env python3
"""Forward+backward step time of the model restatements at the shapes of BASELINE configs 3-5 (GPU box),
with the per-kernel breakdown of one step from the torch profiler.  Not the headline bench (bench.py).
  cfg4  model_multitask.py: batch 4096 x N=50 (variable true sizes), F=81, 12 tasks
  cfg5  model_gin.py layers at D=256 on 10-node ring graphs, batch 20000
  cfg3  sparse.py: 128 molecules x <=50 nodes block-diagonal, F=128 -> 256 x3
With --roofline every C-ABI call of one extra step is timed on its own (tools/abi_roofline.py) and priced with its
algorithmic bytes / flops: the per-entry-point fraction-of-peak table of profiles/r02_*_config_rooflines.json.
--graph additionally captures the step (all of its launches and allocations) in ONE hipGraph and times its replay: the
step without host launch gaps.
usage: python tools/config_bench.py [--roofline] [--graph] [cfg4|cfg5|cfg3 ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import kgcn_oracle as K  # noqa: E402  (graph generators only)
from kgcn_amd import BatchedAdjacency, BatchedCSR, data_util as D, layers, models  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def timed(step, reps=20, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def graphed(step, reps=20):
    """ms per replay of the step captured in one hipGraph (gradients accumulate into static .grad tensors)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def breakdown(step, top=8):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:top]
    return [{"kernel": r.key[:70], "calls": r.count, "us": round(r.device_time_total, 1)} for r in rows]


from config_bench_util import mol_batch  # noqa: E402


def cfg4():
    B, N, F, T = 4096, 50, 81, 12
    sizes, csr = mol_batch(B, N)
    adj = BatchedAdjacency([csr])
    x = torch.randn(B, N, F, device=dev)
    labels = (torch.rand(B, T, device=dev) < 0.5).float()
    mask, ml = torch.ones(B, device=dev), (torch.rand(B, T, device=dev) < 0.8).float()
    en = torch.as_tensor(sizes).to(dev)                  # resident: no host copy inside the step
    model = models.MultitaskGCN(1, T).to(dev)
    model(x, adj, enabled_node_nums=en)

    def step():
        model.zero_grad(set_to_none=True)
        c, _ = models.masked_sigmoid_ce(model(x, adj, enabled_node_nums=en), labels, mask, ml)
        c.backward()
    return B, step


def cfg5():
    B, N, Dm = 20000, 10, 256
    adjs = K.synth_ring_graphs(rng, 2000, N)
    reps = B // 2000
    mats = [a[0] for a in adjs] * reps
    adj = BatchedAdjacency([BatchedCSR.from_coo_list(mats, rows=N, cols=N, device=dev)])
    x = torch.randn(B, N, Dm, device=dev, requires_grad=True)
    gin = [layers.GINAggregate(1).to(dev) for _ in range(2)]
    dn = [layers.GraphDense(Dm, activation="relu").to(dev) for _ in range(4)]      # tf.nn.relu(GraphDense(..)), model_gin.py:45-54
    gather = layers.GraphGather()

    def fwd():
        h, outs = x, []
        for blk in range(2):
            h = gin[blk](h, adj=adj)
            h = dn[2 * blk](h)
            h = dn[2 * blk + 1](h)
            outs.append(gather(h))
        return torch.cat(outs, 1)
    fwd()

    def step():
        for m in gin + dn:
            m.zero_grad(set_to_none=True)
        fwd().sum().backward()
    return B, step


def cfg3():
    nmol, F = 128, 128
    ex, sizes = [], []
    for _ in range(nmol):
        n = int(rng.integers(20, 51))
        idx, val, _ = K.synth_mol_graphs(rng, 1, n, 3)[0][0]
        a = np.zeros((n, n), np.float32)
        a[idx[:, 0], idx[:, 1]] = val
        ex.append(K.sparse_example(a, rng.standard_normal((n, F)).astype(np.float32)))
        sizes.append(n)
    f = K.collate_sparse_examples(ex)
    batch = D.block_diagonal_batch(f["size"][:, 0], f["adj_row"], f["adj_column"], f["adj_values"], f["adj_elem_len"],
                                   f["adj_degrees"], f["feature_row"], f["feature_column"], f["feature_values"],
                                   f["feature_elem_len"], F, max_degree=0, normalize=True, device=dev)
    model = models.SparseGCN(10).to(dev)
    labels = torch.randint(0, 10, (nmol,), device=dev)
    model(batch)

    def step():
        model.zero_grad(set_to_none=True)
        models.sparse_softmax_ce_sum(model(batch), labels).backward()
    return nmol, step


res = {}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rec = None
if "--roofline" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import abi_roofline
    rec = abi_roofline.instrument()
for name in (args or ["cfg4", "cfg5", "cfg3"]):
    graphs, step = {"cfg4": cfg4, "cfg5": cfg5, "cfg3": cfg3}[name]()
    ms = timed(step)
    res[name] = {"graphs_per_step": graphs, "ms_per_step": round(ms, 3), "graphs_per_s": round(graphs / ms * 1e3),
                 "top_kernels": breakdown(step, top=40 if rec else 8)}
    res[name]["kernel_us_sum"] = round(sum(k["us"] for k in res[name]["top_kernels"]), 1)
    if "--graph" in sys.argv:
        res[name]["ms_per_step_hipgraph"] = round(graphed(step), 3)
    if rec:
        rec.calls, rec.on = [], True
        step()
        torch.cuda.synchronize()
        rec.on = False
        rows = rec.rows()
        res[name]["abi_calls"] = rows
        res[name]["abi_us_sum"] = round(sum(r["us"] for r in rows), 1)
        res[name]["abi_unpriced"] = sorted(rec.other)
print(json.dumps(res, indent=1))
