"""Two data-parallel ranks with the REAL kernels on one GPU (both on cuda:0, gloo backend -- RCCL refuses two ranks on one
device): the sharded training step of kgcn_amd.train / kgcn_amd.parallel (contiguous shards of unequal size padded with dummy
graphs, local-mean gradients weighted B_r / B, one all-reduce of the optimiser's flat gradient buffer, fused TF-Adam update)
must follow the single-process step on the whole batch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

_RANK_SCRIPT = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
root, port, rank, world, outdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.path.insert(0, root)
from kgcn_amd import data_util as D, models, train, parallel
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
dev = torch.device("cuda:0")
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
raw = np.load(os.path.join(root, "tests", "golden", "g1_synthetic_raw.npz"))
chans, _ = D.build_adjs({"dense_adj": raw["dense_adj"].astype(np.int64), "max_node_num": 10})
ds = D.DeviceGraphDataset(chans, raw["feature"], device=dev)
labels = raw["label"].astype(np.float32)
torch.manual_seed(0)
model = models.GCN(1).to(dev)
adj0, x0 = ds.batch(np.arange(8), 8)
model(x0, adj0)                                            # Keras-style build: the same initial weights on every rank
opt = train.TFAdam(model.parameters(), lr=0.01)
bucket = parallel.GradBucket(list(model.parameters()), flat=opt.flat) if world > 1 else None
GLOBAL = 47                                                # graphs per global batch: shards of 24 and 23
costs = []
for step in range(4):
    idx = (np.arange(GLOBAL) * 3 + 11 * step) % 200
    lo, hi = parallel.shard_range(GLOBAL, rank, world)
    pad = 48 // world                                      # 47 graphs + dummies: one padded batch of 48, or two of 24
    mine = idx[lo:hi]
    adj, x = ds.batch(mine, pad)
    lab = torch.zeros((pad, 2), device=dev); lab[:len(mine)] = torch.from_numpy(labels[mine]).to(dev)
    mask = torch.zeros(pad, device=dev); mask[:len(mine)] = 1
    w = parallel.shard_weight(pad, pad * world) if world > 1 else None
    cs, _ = train.train_step(model, opt, models.masked_softmax_ce, x, adj, lab, mask, bucket=bucket, shard_weight=w)
    if world > 1:
        t = torch.tensor([cs], dtype=torch.float64)
        dist.all_reduce(t)
        cs = float(t)
    costs.append(cs)
torch.cuda.synchronize()
if rank == 0:
    np.savez(os.path.join(outdir, "world%d.npz" % world), costs=np.array(costs),
             **{"p%d" % i: p.detach().cpu().numpy() for i, p in enumerate(model.parameters())})
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_ranks_follow_the_single_process_step(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), ROOT, "0", "0", "1", str(tmp_path)], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    for attempt in range(3):
        port = str(_free_port())
        procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(k), "2", str(tmp_path)], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env) for k in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        if all(p.returncode == 0 for p in procs):
            break
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-2000:] for o in outs)
    one, two = np.load(tmp_path / "world1.npz"), np.load(tmp_path / "world2.npz")
    # the single process sees 47 graphs + 1 dummy in ONE padded batch of 48; the two ranks 24 and 23 + 1 dummy: the same padded mean
    np.testing.assert_allclose(two["costs"], one["costs"], rtol=2e-5)
    assert one["costs"][-1] < one["costs"][0]
    for k in one.files:
        if k.startswith("p"):
            np.testing.assert_allclose(two[k], one[k], rtol=2e-4, atol=2e-5, err_msg=k)
