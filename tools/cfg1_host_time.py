#!/usr/bin/env python3
"""development: host / device split of bench.py's cfg1 step (batch assembly launched eagerly + hipGraph replay)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = bench.build_parser().parse_args(["--config", "cfg1", "--graphs", "20000", "--batch", sys.argv[1] if len(sys.argv) > 1 else "4096"])
ctx = bench.Ctx(args)
wl = bench.Cfg1(args, ctx)
for _ in range(10):
    wl.step()
torch.cuda.synchronize()
def t(fn, n=100, sync_each=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("step (pipelined)      %.3f ms" % t(wl.step))
print("step (sync each)      %.3f ms" % t(wl.step, sync_each=True))
print("next_batch sync each  %.3f ms" % t(wl.next_batch, sync_each=True))
print("next_batch pipelined  %.3f ms" % t(wl.next_batch))
print("replay sync each      %.3f ms" % t(wl.graph_step.replay, sync_each=True))
print("replay pipelined      %.3f ms" % t(wl.graph_step.replay))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): wl.next_batch()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
