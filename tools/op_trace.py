#!/usr/bin/env python3
"""development: which aten ops / autograd nodes launch the small torch kernels of a bench.py step (one eager step under
torch.profiler).  usage: op_trace.py <bench args...>"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
args = bench.build_parser().parse_args(sys.argv[1:] + ["--eager"])
ctx = bench.Ctx(args)
wl = {"cfg1": bench.Cfg1, "cfg4": bench.Cfg4, "cfg5": bench.Cfg5, "cfg3": bench.Cfg3}[args.config](args, ctx)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    wl.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
