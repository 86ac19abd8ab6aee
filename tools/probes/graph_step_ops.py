"""which aten ops run inside the captured train step (GraphedTrainStep._eager on a side stream, as during warm-up)"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
args = bench.build_parser().parse_args(sys.argv[1:])
ctx = bench.Ctx(args)
wl = {"cfg1": bench.Cfg1, "cfg4": bench.Cfg4, "cfg5": bench.Cfg5, "cfg3": bench.Cfg3}[args.config](args, ctx)
gs = wl.graph_step
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    gs._eager()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") or "Memcpy" in e.key or "AccumulateGrad" in e.key]
for e in sorted(rows, key=lambda e: -e.count)[:40]:
    print("%-40s n=%-3d shapes=%s  cuda=%.1fus" % (e.key[:40], e.count, str(e.input_shapes)[:90], e.device_time_total))
