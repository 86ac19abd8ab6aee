"""Synthetic molecule batches shared by the tools/ benches."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kgcn_oracle as K  # noqa: E402  (graph generators only)
from kgcn_amd import BatchedCSR  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def mol_batch(B, N, extra=2):
    sizes = rng.integers(5, N + 1, size=B)
    g, r, c = [], [], []
    for b, n in enumerate(sizes):
        idx = K.synth_mol_graphs(rng, 1, int(n), extra)[0][0][0]
        g.append(np.full(len(idx), b)); r.append(idx[:, 0]); c.append(idx[:, 1])
    g, r, c = np.concatenate(g), np.concatenate(r), np.concatenate(c)
    deg = np.bincount(g * N + c, minlength=B * N).astype(np.float32)
    deg[deg == 0] = 1
    rs = (1.0 / np.sqrt(deg)).astype(np.float32)
    val = rs[g * N + r] * rs[g * N + c]
    return sizes, BatchedCSR.from_arrays(g, r, c, val, B, N, N, device=dev)
