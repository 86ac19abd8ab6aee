#!/usr/bin/env python3
"""Reference-API feed (lists of per-graph COO triples) -> A, A^T and both row-padded containers: the device packer
(kgcn_coo_pack_f32 / kgcn_csr_pad4) against the numpy packer, host time per batch (GPU box)."""
import os
import sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kgcn_oracle as K
from kgcn_amd import BatchedCSR, batched_csr
rng = np.random.default_rng(0)
adjs = K.synth_mol_graphs(rng, 4096, 32, 3)
mats = [a[0] for a in adjs]
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full(b):
    b.transpose(); b.padded4(); b.transpose().padded4(); return b
dev_ms = t(lambda: full(BatchedCSR.from_coo_list(mats, rows=32, cols=32, device="cuda")))
batched_csr.DEVICE_PACK_MIN_NNZ = 10 ** 12
host_ms = t(lambda: full(BatchedCSR.from_coo_list(mats, rows=32, cols=32, device="cuda")))
print("4096 graphs x 100 entries: A, A^T and both row-padded copies from COO lists: device packer %.1f ms, numpy packer %.1f ms" % (dev_ms, host_ms))
