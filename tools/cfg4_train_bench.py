#!/usr/bin/env python3
"""BASELINE config 4 end to end: Tox21-shaped multitask training (example_model/model_multitask.py,
example_config/multitask.json shape: 12 tasks, N = 50 padded with variable true sizes, F = 81, masked labels),
dataset resident in HBM, batch assembled on the device, forward + masked sigmoid CE + backward + TF-Adam --
eager launches and the hipGraph-captured step (kgcn_amd.train.GraphedTrainStep).
Synthetic molecules (random tree + extra edges + self loops on the first `size` nodes, Kipf-normalised values).

Data parallel: launched through torch.distributed.run (one rank per GPU, RCCL) every rank holds its own `graphs`
molecules and steps on `batch` of them (weak scaling); the flat gradient bucket is all-reduced between backward and
the Adam update -- inside the captured graph in graph mode.  Time = max over ranks between two barriers; rank 0
prints the JSON.
usage: python tools/cfg4_train_bench.py [graphs=200000] [batch=4096] [steps=30]
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
              tools/cfg4_train_bench.py [graphs] [batch] [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgcn_amd import data_util as D, models, parallel, train  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 30
N, F, T = 50, 81, 12
WORLD = int(os.environ.get("WORLD_SIZE", "1"))
RANK = int(os.environ.get("RANK", "0"))
LOCAL = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(LOCAL)
dev = torch.device("cuda", LOCAL)
DP = "RANK" in os.environ                               # under torch.distributed.run: also with ONE rank (exercises the
if DP:                                                  # captured all-reduce on a 1-GPU box)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=RANK, world_size=WORLD, device_id=dev)
rng = np.random.default_rng(4 + RANK)                   # every rank its own molecules

t_gen = time.perf_counter()
sizes = rng.integers(5, N + 1, size=G)
A = np.zeros((G, N, N), np.bool_)
ar = np.arange(G)
for i in range(1, N):                                   # random tree over the first `size` nodes of every graph
    act = sizes > i
    j = (rng.random(G) * i).astype(np.int64)
    A[ar[act], i, j[act]] = True
    A[ar[act], j[act], i] = True
for _ in range(2):                                      # two extra edges per molecule
    i = (rng.random(G) * sizes).astype(np.int64)
    j = (rng.random(G) * sizes).astype(np.int64)
    A[ar, i, j] = True
    A[ar, j, i] = True
node = np.arange(N)
valid = node[None, :] < sizes[:, None]
A[:, node, node] = valid                                # self loops on real nodes only
g, r, c = np.nonzero(A)
del A
chan = D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), G, N)
chan = D.normalize_adj(chan)
feats = (rng.standard_normal((G, N, F)).astype(np.float32)) * valid[:, :, None]
labels = (rng.random((G, T)) < 0.3).astype(np.float32)
mask_label = (rng.random((G, T)) < 0.8).astype(np.float32)
gen_s = time.perf_counter() - t_gen

ds = D.DeviceGraphDataset([chan], feats, device=dev)
lab_d, ml_d = torch.from_numpy(labels).to(dev), torch.from_numpy(mask_label).to(dev)
sizes_d = torch.from_numpy(sizes).to(dev)
torch.manual_seed(0)                                    # the same initial weights on every rank
model = models.MultitaskGCN(1, T).to(dev)
adj0, x0 = ds.batch(np.arange(B), B)
model(x0, adj0, enabled_node_nums=sizes_d[:B])
del adj0, x0
mask = torch.ones(B, device=dev)
weight = parallel.shard_weight(B, B * WORLD) if DP else None


def barrier():
    torch.cuda.synchronize()
    if DP:
        dist.barrier()
        torch.cuda.synchronize()


def timed(run):
    run(3)
    barrier()
    t0 = time.perf_counter()
    last = run(STEPS)
    barrier()
    dt = time.perf_counter() - t0
    if DP:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, float(last.detach()) if torch.is_tensor(last) else float(last)


def batches(steps):
    perm = rng.permutation(G)
    return [perm[(s * B) % (G - B):(s * B) % (G - B) + B] for s in range(steps)]


# ---- eager launches -------------------------------------------------------------------------------------------
opt = train.TFAdam(model.parameters(), lr=1e-3)
bucket = parallel.GradBucket(list(model.parameters())) if DP else None


def run_eager(steps):
    tot = 0.0
    for idx in batches(steps):
        it = torch.from_numpy(idx).to(dev)
        adj, x = ds.batch(idx, B)
        opt.zero_grad()
        logits = model(x, adj, enabled_node_nums=sizes_d[it])
        cost_opt, cost_sum = models.masked_sigmoid_ce(logits, lab_d[it], mask, ml_d[it])
        cost_opt.backward()
        if bucket is not None:
            bucket.all_reduce_mean(weight=weight)
        opt.step()
        tot = cost_sum
    return tot


dt_e, last_e = timed(run_eager)

# ---- one hipGraph per step --------------------------------------------------------------------------------------
opt_g = train.TFAdam(model.parameters(), lr=1e-3, capturable=True)
bucket_g = parallel.GradBucket(list(model.parameters())) if DP else None
sb = ds.static_batch(B)
sb.load(np.arange(B))
lab_s, ml_s = torch.zeros((B, T), device=dev), torch.zeros((B, T), device=dev)
en_s = torch.zeros(B, device=dev, dtype=sizes_d.dtype)
step = train.GraphedTrainStep(model, opt_g, lambda lg, lb, mk: models.masked_sigmoid_ce(lg, lb, mk, ml_s), sb, lab_s, mask,
                              bucket=bucket_g, shard_weight=weight, enabled_node_nums=en_s)


def run_graph(steps):
    tot = 0.0
    for idx in batches(steps):
        it = torch.from_numpy(idx).to(dev)
        sb.load(idx)
        lab_s.copy_(lab_d[it]); ml_s.copy_(ml_d[it]); en_s.copy_(sizes_d[it])
        tot, _ = step.replay()
    return tot


dt_g, last_g = timed(run_graph)


def replay_only(steps):
    tot = 0.0
    for _ in range(steps):
        tot, _ = step.replay()
    return tot


dt_r, _ = timed(replay_only)

if RANK == 0:
    def line(dt):
        return {"ms_per_step": round(dt / STEPS * 1e3, 4), "graphs_per_s": round(B * WORLD * STEPS / dt)}
    print(json.dumps({"config": "cfg4: model_multitask.py, %d graphs resident per GPU, batch %d per GPU, N=%d (true sizes 5..50), "
                                "F=%d, %d tasks" % (G, B, N, F, T),
                      "n_gpus": WORLD, "scaling": "weak", "steps": STEPS,
                      "eager": line(dt_e), "hipgraph": line(dt_g), "hipgraph_replay_only": line(dt_r),
                      "collective": None if not DP else {"backend": "nccl (RCCL)", "ranks": dist.get_world_size(),
                                                             "bucket_floats": int(sum(bucket_g.sizes)),
                                                             "in_graph": True},
                      "dataset_bytes_hbm": int(feats.nbytes + 8 * g.shape[0] + 4 * (G * N + 1)),
                      "host_generation_s": round(gen_s, 2), "final_cost_sum": {"eager": last_e, "hipgraph": last_g}}, indent=1))
if DP:
    dist.destroy_process_group()
