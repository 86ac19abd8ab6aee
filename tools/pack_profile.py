import os, sys, time, numpy as np, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
from oracle import kgcn_oracle as K
from kgcn_amd import BatchedCSR, batched_csr
rng = np.random.default_rng(0)
adjs = K.synth_mol_graphs(rng, 4096, 32, 3)
mats = [a[0] for a in adjs]
def full(b):
    b.transpose(); b.padded4(); b.transpose().padded4(); return b
f = lambda: full(BatchedCSR.from_coo_list(mats, rows=32, cols=32, device="cuda"))
for _ in range(3): f()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): f()
torch.cuda.synchronize(); print("ms", (time.perf_counter()-t0)/10*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): f()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
