#!/usr/bin/env python3
"""bench.py -- graphs/sec of GraphConv forward+backward on synthetic molecular graphs.

Metric (BASELINE.json): "graphs/sec GraphConv fwd+bwd, 32-node mol graphs x64 feat".
Workload = BASELINE config 2: 100,000 random 32-node graphs per GPU (random
spanning tree + 3 extra edges, symmetrised, + self loops => nnz = 100 exactly, values 1.0), 64-dim
features, one adjacency channel, kernel [64,64] glorot-uniform.  One "step" = one pass of the hot
path over that batch: kgcn_amd.layers.GraphConv forward, then backward producing dX, dW, dbias
(+ for N > 1 one RCCL all-reduce of the flat [dW, dbias] bucket).  Inputs are resident in HBM
before the timed region.  Weak scaling: every rank owns its own 100k graphs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graphs G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the fused backward) against
the 8 TB/s HBM peak using the ALGORITHMIC bytes of DESIGN.md, with median / p10 / p90 of the per-launch
HIP-event times; `roofline.spmm_kernel` prices the batched SpMM alone (kgcn_bspmm_f32 forward
and adjoint on the same batch, 17,316 B/graph -- the kernel the north-star 60 % target is stated on), timed
after the K steps; `cpu_baseline` times the C restatement of the reference algorithm (oracle/kgcn_ref.c,
OpenMP over graphs) on the host cores on a bounded sample of the same workload, next to a reference-SHAPED
leg (per-graph scipy CSR @ (X W + b) loop, the op structure of kgcn/layers.py:107-116).

--scaling weak (default): every rank owns --graphs graphs.  --scaling strong: --graphs is the GLOBAL batch,
sharded contiguously over the ranks (kgcn_amd.parallel.shard_range); gradients combine with the shard weights.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES = 32
FEAT = 64
EXTRA_EDGES = 3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# ---------------------------------------------------------------------------------------------
# workload (vectorised version of oracle.kgcn_oracle.synth_mol_graphs; same distribution)
# ---------------------------------------------------------------------------------------------
def gen_mol_graphs(T, n=N_NODES, extra=EXTRA_EDGES, seed=1234):
    """Returns flat COO (graph, row, col) in row-major order; nnz = T * (2*(n-1+extra) + n)."""
    rng = np.random.default_rng(seed)
    A = np.zeros((T, n, n), np.bool_)
    perm = rng.permuted(np.tile(np.arange(n), (T, 1)), axis=1)
    ar = np.arange(T)
    for i in range(1, n):                                   # random spanning tree
        j = perm[ar, rng.integers(0, i, size=T)]
        A[ar, perm[:, i], j] = True
        A[ar, j, perm[:, i]] = True
    need = np.full(T, extra)
    while (need > 0).any():                                 # extra edges, rejection sampled
        act = np.nonzero(need > 0)[0]
        i = rng.integers(0, n, size=act.size)
        j = rng.integers(0, n, size=act.size)
        ok = (i != j) & ~A[act, i, j]
        a = act[ok]
        A[a, i[ok], j[ok]] = True
        A[a, j[ok], i[ok]] = True
        need[a] -= 1
    A[:, np.arange(n), np.arange(n)] = True                 # self loops
    g, r, c = np.nonzero(A)
    return g.astype(np.int64), r.astype(np.int64), c.astype(np.int64)


def kipf_values(g, r, c, T, n):
    """normalize_adj (kgcn/data_util.py:125-140) on binary symmetric adjacency, float32."""
    deg = np.bincount(g * n + c, minlength=T * n).astype(np.float32)
    deg[deg == 0] = 1
    recip = (1.0 / np.sqrt(deg)).astype(np.float32)
    return (np.float32(1.0) * recip[g * n + r]) * recip[g * n + c]


def make_cfg2(T, device, seed=1234, normalize=False):
    import torch
    from kgcn_amd import BatchedCSR
    g, r, c = gen_mol_graphs(T, seed=seed)
    val = kipf_values(g, r, c, T, N_NODES) if normalize else np.ones(g.shape[0], np.float32)
    csr = BatchedCSR.from_arrays(g, r, c, val, T, N_NODES, N_NODES, device=device)
    csr.transpose()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    x = torch.randn((T, N_NODES, FEAT), device=device, dtype=torch.float32, generator=gen)
    grad = torch.randn((T, N_NODES, FEAT), device=device, dtype=torch.float32, generator=gen)
    wrng = np.random.default_rng(4321)                      # identical weights on every rank
    lim = np.sqrt(6.0 / (FEAT + FEAT))
    w = torch.as_tensor(wrng.uniform(-lim, lim, size=(FEAT, FEAT)).astype(np.float32), device=device)
    bias = torch.zeros((1, FEAT), device=device, dtype=torch.float32)
    off = np.zeros(T + 1, np.int64)
    np.cumsum(np.bincount(g, minlength=T), out=off[1:])
    idx = np.stack([r, c], axis=1).astype(np.int32)

    def adjs_of(pick):
        return [[(idx[off[t]:off[t + 1]], val[off[t]:off[t + 1]], [N_NODES, N_NODES])] for t in pick]

    return dict(csr=csr, x=x, g=grad, w=w, bias=bias, off=off, idx=idx, val=val, adjs_of=adjs_of,
                nnz_per_graph=float(g.shape[0]) / T)


# ---------------------------------------------------------------------------------------------
# algorithmic bytes (SURVEY 8d / DESIGN.md): fp32 values, int32 indices, W/bias on chip
# ---------------------------------------------------------------------------------------------
def algorithmic_bytes(n, din, dout, nnz):
    csr = 4 * (n + 1) + 8 * nnz
    fwd = 4 * n * din + csr + 4 * n * dout
    bwd = 4 * n * dout + csr + 4 * n * din + 4 * n * din     # read g, CSR, read x, write dx
    return dict(csr=csr, fwd=fwd, bwd=bwd, layer=fwd + bwd)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the C restatement (oracle/kgcn_ref.c) on the host cores -- reported, never shipped
# ---------------------------------------------------------------------------------------------
def cpu_baseline(wl, budget_s=12.0, sample=20000):
    from oracle import ref_c
    T = min(sample, wl["csr"].num_graphs)
    off = wl["off"][:T + 1].copy()
    nnz = int(off[-1])
    idx = np.ascontiguousarray(wl["idx"][:nnz])
    val = np.ascontiguousarray(wl["val"][:nnz])
    x = wl["x"][:T].detach().cpu().numpy()
    g = wl["g"][:T].detach().cpu().numpy()
    w = wl["w"].cpu().numpy()
    b = wl["bias"].cpu().numpy()
    threads = ref_c.max_threads()
    ref_c.graphconv_fwd(off, idx, val, x[:256], w, b)       # warm up (page in, spawn threads)
    reps, t0 = 0, time.perf_counter()
    while True:
        ref_c.graphconv_fwd(off, idx, val, x, w, b)
        ref_c.graphconv_bwd(off, idx, val, x, w, g)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s and reps >= 3:
            break
    res = {"value": T * reps / el, "unit": "graphs/sec", "cores": threads, "kind": "port",
           "sample": "%d graphs x %d passes of GraphConv fwd+bwd (oracle/kgcn_ref.c, OpenMP %d "
                     "threads, fp32), %.1f s" % (T, reps, threads, el)}
    # reference-shaped leg: one scipy CSR SpMM and one tiny GEMM per graph, forward and the a-6 / a-7
    # backward, single thread -- the op structure of kgcn/layers.py:107-116, not its TF executor
    import scipy.sparse as sp
    Ts = min(2000, T)
    mats = [sp.csr_matrix((val[off[t]:off[t + 1]], (idx[off[t]:off[t + 1], 0], idx[off[t]:off[t + 1], 1])),
                          shape=(N_NODES, N_NODES)) for t in range(Ts)]
    t0, done = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        dw = np.zeros_like(w)
        db = np.zeros((1, w.shape[1]), np.float32)
        for t in range(Ts):
            out = mats[t] @ (x[t] @ w + b)                                        # noqa: F841
            dfw = mats[t].T @ g[t]
            dx = dfw @ w.T                                                        # noqa: F841
            dw += x[t].T @ dfw
            db += dfw.sum(0, keepdims=True)
        done += Ts
    el2 = time.perf_counter() - t0
    res["reference_shaped"] = {"value": done / el2, "unit": "graphs/sec", "cores": 1,
                               "sample": "%d graphs, per-graph scipy.sparse CSR @ (X W + b) + backward, numpy fp32, "
                                         "1 thread, %.1f s" % (done, el2)}
    return res


# ---------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graphs", type=int, default=100_000, help="graphs per GPU per step")
    ap.add_argument("--normalize", action="store_true", help="Kipf-normalised adjacency values")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="dense GEMM + Bspmm kernels instead of the fused layer")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --graphs per GPU; strong: --graphs in total, sharded over the GPUs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from kgcn_amd import layers
    from kgcn_amd.parallel import GradBucket, shard_range, shard_weight

    if args.scaling == "strong":
        lo, hi = shard_range(args.graphs, rank, world)          # contiguous shard of ONE global batch
        T, T_global = hi - lo, args.graphs
    else:
        T, T_global = args.graphs, args.graphs * world
    weight = shard_weight(T, T_global) if world > 1 else None
    wl = make_cfg2(T, device, seed=1234 + rank, normalize=args.normalize)
    csr = wl["csr"]
    layer = layers.GraphConv(FEAT, 1).to(device)
    layer.build((T, N_NODES, FEAT), device)
    with torch.no_grad():
        layer.w[0].copy_(wl["w"])
        layer.bias[0].copy_(wl["bias"])
    if args.unfused:
        layers.enabled_batched = True
    x = wl["x"].requires_grad_(True)
    g = wl["g"]
    bucket = GradBucket(list(layer.parameters())) if world > 1 else None

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(events=None):
        x.grad = None
        for p in layer.parameters():
            p.grad = None
        if events:
            events[0].record()
        out = layer(x, adj=csr)
        if events:
            events[1].record()
        out.backward(g)
        if events:
            events[2].record()
        if bucket is not None:
            bucket.all_reduce_mean(weight=weight)

    # setup: prime the caching allocator, the lazily built A^T / row-padded containers, the LDS attributes
    # and the clocks (the GPU idles at 107 MHz and needs some tens of milliseconds of load to reach its
    # sustained state, profiles/r01_h_power_clocks.txt) with untimed passes -- part of initialisation, like
    # data generation -- then the W warm-up steps of the contract
    for _ in range(25):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the batched SpMM alone (the kernel of the north-star 60 % target): forward and adjoint launches on the same batch,
    # after the timed region, clocks still in their sustained state
    spmm = None
    if rank == 0 and not args.unfused:
        from kgcn_amd import ops
        x2d, g2d = wl["x"].detach().reshape(T * N_NODES, FEAT), g.reshape(T * N_NODES, FEAT)
        o2d = torch.empty_like(x2d)
        csr_t = csr.transpose()
        sev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(max(args.steps, 20))]
        for _ in range(5):
            ops.bspmm_raw(csr, x2d, FEAT, o2d)
            ops.bspmm_raw(csr_t, g2d, FEAT, o2d)
        for e in sev:
            e[0].record()
            ops.bspmm_raw(csr, x2d, FEAT, o2d)
            e[1].record()
            ops.bspmm_raw(csr_t, g2d, FEAT, o2d)
            e[2].record()
        torch.cuda.synchronize()
        spmm = ([e[0].elapsed_time(e[1]) for e in sev], [e[1].elapsed_time(e[2]) for e in sev])

    def stats(ms):
        ms = sorted(ms)
        n = len(ms)
        return {"median_ms": ms[n // 2], "p10_ms": ms[n // 10], "p90_ms": ms[(9 * n) // 10], "mean_ms": float(np.mean(ms))}

    if rank == 0:
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_cfg2.json")
        if os.path.exists(tpath) and not args.unfused:
            tj = json.load(open(tpath))
            if tj.get("graphs_per_launch") == T:      # PMC-measured HBM bytes of the same launch shape
                traffic = {k: v.get("bytes") for k, v in tj.items() if isinstance(v, dict)}
        fwd_st = stats([e[0].elapsed_time(e[1]) for e in ev])
        bwd_st = stats([e[1].elapsed_time(e[2]) for e in ev])
        fwd_ms, bwd_ms = fwd_st["mean_ms"], bwd_st["mean_ms"]
        ab = algorithmic_bytes(N_NODES, FEAT, FEAT, wl["nnz_per_graph"])
        bwd_gbs = ab["bwd"] * T / (bwd_ms * 1e-3) / 1e9
        fwd_gbs = ab["fwd"] * T / (fwd_ms * 1e-3) / 1e9
        traffic_bwd = traffic.get("graphconv_bwd_planes_kernel")
        traffic_fwd = traffic.get("graphconv_fwd_full_kernel")
        spmm_entry = None
        if spmm is not None:
            sb = 2 * 4 * N_NODES * FEAT + ab["csr"]
            sf, sa = stats(spmm[0]), stats(spmm[1])
            spmm_entry = {"kernel": "spmm_tile_kernel (kgcn_bspmm_f32: Bspmm / Bspmdt / Bconv)", "bound": "hbm",
                          "algorithmic_bytes_per_graph": sb, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "forward": dict(sf, achieved=sb * T / (sf["median_ms"] * 1e-3) / 1e9,
                                          frac=sb * T / (sf["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS),
                          "adjoint": dict(sa, achieved=sb * T / (sa["median_ms"] * 1e-3) / 1e9,
                                          frac=sb * T / (sa["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS),
                          "traffic": traffic.get("spmm_tile_kernel")}
        res = {
            "metric": "graphs/sec GraphConv fwd+bwd, 32-node mol graphs x64 feat",
            "value": T_global * args.steps / elapsed,
            "unit": "graphs/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: %d random 32-node graphs per GPU (tree+3 edges+self loops, "
                                   "nnz=100), 64-dim features, 1 adjacency channel, GraphConv "
                                   "fwd+bwd (dX,dW,dbias)%s" % (T, ", unfused kernels" if args.unfused else ""),
                       "graphs_per_gpu": T, "graphs_global": T_global, "n_nodes": N_NODES, "din": FEAT, "dout": FEAT,
                       "nnz_per_graph": wl["nnz_per_graph"], "parallelism": "dp%d" % world,
                       "collective": None if world == 1 else "one RCCL all-reduce of the flat [dW, dbias] bucket "
                                                            "(%d floats) per step over %d ranks (backend %s)"
                                                            % (bucket.total, dist.get_world_size(), dist.get_backend()),
                       "adjacency_values": "kipf" if args.normalize else "ones"},
            "roofline": {"bound": "hbm",
                         "kernel": "dense_wgrad+bspmm (unfused)" if args.unfused else
                                   "graphconv_bwd_planes_kernel (+1 reduce_partials launch, ~5 us, in the event bracket)",
                         "achieved": bwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bwd_gbs / HBM_PEAK_GBS, "traffic": traffic_bwd,
                         "launch_ms": bwd_st,
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC (profiles/traffic_cfg2.json); "
                                         "algorithmic bytes per launch = %d" % int(ab["bwd"] * T),
                         "algorithmic_bytes_per_graph": ab["bwd"], "avg_launch_ms": bwd_ms,
                         "fwd_kernel": {"achieved": fwd_gbs, "frac": fwd_gbs / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_graph": ab["fwd"], "avg_launch_ms": fwd_ms,
                                        "launch_ms": fwd_st, "traffic": traffic_fwd},
                         "spmm_kernel": spmm_entry,
                         "layer_frac_of_hbm_peak": ab["layer"] * T / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
