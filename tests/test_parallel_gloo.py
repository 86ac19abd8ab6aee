"""World-size-2 data-parallel test on CPU (gloo): the shard + flat-bucket all-reduce logic of
kgcn_amd.parallel.  Local gradients come from the ORACLE here (the product kernels need a GPU);
what is tested is that mean-reduced shard gradients equal the single-process gradients of the
concatenated batch, and that the shards tile the batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kgcn_amd.parallel import GradBucket, shard_range
    from oracle import kgcn_oracle as K
    rng = np.random.default_rng(0)                       # same data on every rank
    T = 24
    adjs = K.synth_mol_graphs(rng, T, 16, 2)
    x = rng.standard_normal((T, 16, 8))
    w = [rng.standard_normal((8, 12))]
    b = [rng.standard_normal((1, 12))]
    g = rng.standard_normal((T, 16, 12))
    lo, hi = shard_range(T, rank, world)
    _, dw, db = K.graphconv_bwd(x[lo:hi], adjs[lo:hi], w, b, g[lo:hi])
    pw = torch.nn.Parameter(torch.tensor(w[0], dtype=torch.float32))
    pb = torch.nn.Parameter(torch.tensor(b[0], dtype=torch.float32))
    pw.grad = torch.tensor(dw[0], dtype=torch.float32)
    pb.grad = torch.tensor(db[0], dtype=torch.float32)
    GradBucket([pw, pb]).all_reduce_mean()
    _, dw_all, db_all = K.graphconv_bwd(x, adjs, w, b, g)
    ok = (np.allclose(pw.grad.numpy() * world, dw_all[0], rtol=1e-5, atol=1e-4)
          and np.allclose(pb.grad.numpy() * world, db_all[0], rtol=1e-5, atol=1e-4))
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d" % (int(ok), lo, hi))
    dist.destroy_process_group()


def _worker_run(rank, world, port, outdir):
    """gradients that already lie as consecutive slices of ONE buffer (what the fused GraphConv backward leaves: [dW | dbias]):
    the exchange reduces that buffer in place -- no pack, no unpack -- for equal and for weighted shards"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kgcn_amd.parallel import GradBucket
    ok = True
    for weight in (None, 0.25 if rank == 0 else 0.75):
        pw, pb = torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(1, 7))
        buf = torch.arange(42, dtype=torch.float32) * (rank + 1)
        pw.grad, pb.grad = buf[:35].view(5, 7), buf[35:].view(1, 7)
        bucket = GradBucket([pw, pb])
        assert bucket._contiguous_run([pw.grad, pb.grad]) is not None
        out = bucket.all_reduce_mean(weight=weight)
        base = torch.arange(42, dtype=torch.float32)
        want = base * 1.5 if weight is None else base * (0.25 * 1 + 0.75 * 2)
        ok = ok and out.data_ptr() == buf.data_ptr() and bucket._flat is None            # reduced where it lay
        ok = ok and torch.allclose(buf, want) and torch.allclose(pw.grad.reshape(-1), want[:35]) and torch.allclose(pb.grad.reshape(-1), want[35:])
    # separate allocations fall back to the packed path
    pw, pb = torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(1, 7))
    pw.grad, pb.grad = torch.ones(5, 7) * (rank + 1), torch.ones(1, 7) * (rank + 1)
    b2 = GradBucket([pw, pb])
    ok = ok and b2._contiguous_run([pw.grad, pb.grad]) is None
    b2.all_reduce_mean()
    ok = ok and torch.allclose(pw.grad, torch.full((5, 7), 1.5))
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write(str(int(ok)))
    dist.destroy_process_group()


def test_dp_contiguous_gradient_run_is_reduced_in_place_world2(tmp_path):
    mp.spawn(_worker_run, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / ("rank%d.txt" % r)).read() == "1"


def test_dp_gradient_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    spans = []
    for r in range(world):
        ok, lo, hi = open(tmp_path / ("rank%d.txt" % r)).read().split()
        assert ok == "1", "rank %d: reduced gradients differ from the full-batch gradients" % r
        spans.append((int(lo), int(hi)))
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 24


def _worker_unequal(rank, world, port, outdir):
    """Unequal shards with padded dummy graphs (quirk Q5): the loss is a MEAN over the padded batch, so rank r's
    local-mean gradients enter with weight B_r / B; the weighted bucket must equal the single-process gradients of
    the mean loss over the whole padded batch."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kgcn_amd.parallel import GradBucket, shard_range, shard_weight
    from oracle import kgcn_oracle as K
    rng = np.random.default_rng(1)
    real, T = 20, 25                                          # 20 real graphs padded to a batch of 25 (feed.py:123-126)
    adjs = K.synth_mol_graphs(rng, real, 16, 2)
    empty = [(np.zeros((0, 2), np.int32), np.zeros(0, np.float32), [16, 16])]
    adjs = adjs + [empty for _ in range(T - real)]
    x = np.concatenate([rng.standard_normal((real, 16, 8)), np.zeros((T - real, 16, 8))])
    w = [rng.standard_normal((8, 12))]
    b = [rng.standard_normal((1, 12))]
    g_sum = rng.standard_normal((T, 16, 12))                  # d(sum loss)/d out; the mean loss divides by the batch
    lo, hi = shard_range(T, rank, world)                      # 13 / 12 graphs
    _, dw, db = K.graphconv_bwd(x[lo:hi], adjs[lo:hi], w, b, g_sum[lo:hi] / (hi - lo))     # local mean loss
    pw = torch.nn.Parameter(torch.tensor(w[0], dtype=torch.float32))
    pb = torch.nn.Parameter(torch.tensor(b[0], dtype=torch.float32))
    pw.grad = torch.tensor(dw[0], dtype=torch.float32)
    pb.grad = torch.tensor(db[0], dtype=torch.float32)
    GradBucket([pw, pb]).all_reduce_mean(weight=shard_weight(hi - lo, T))
    _, dw_all, db_all = K.graphconv_bwd(x, adjs, w, b, g_sum / T)                           # global mean loss
    ok = (np.allclose(pw.grad.numpy(), dw_all[0], rtol=1e-5, atol=1e-5)
          and np.allclose(pb.grad.numpy(), db_all[0], rtol=1e-5, atol=1e-5))
    # the plain mean is NOT the same thing for 13 / 12 graphs
    plain = (hi - lo) != T // world or T % world == 0
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d" % (int(ok), lo, hi, int(plain)))
    dist.destroy_process_group()


def test_dp_unequal_shards_weighted_mean_world2(tmp_path):
    world = 2
    mp.spawn(_worker_unequal, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sizes = []
    for r in range(world):
        ok, lo, hi, _ = open(tmp_path / ("rank%d.txt" % r)).read().split()
        assert ok == "1", "rank %d: weighted bucket differs from the gradients of the global mean loss" % r
        sizes.append(int(hi) - int(lo))
    assert sorted(sizes) == [12, 13]


def test_empty_parameter_lists_are_refused():
    from kgcn_amd.parallel import GradBucket
    from kgcn_amd.train import TFAdam
    import pytest
    with pytest.raises(ValueError, match="first forward"):
        GradBucket([])
    with pytest.raises(ValueError, match="first forward"):
        TFAdam([])


def test_shard_range_tiles_any_batch():
    from kgcn_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 100_000, 100_003):
        for w in (1, 2, 4, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
