#!/usr/bin/env python
"""Which Python line issues each ATen operator of a training step (every one of them is a kernel launch that is not
libkgcn_hip.so's): one eager step of a bench configuration under a TorchDispatchMode; prints operator, count and the innermost
frames inside this repository.  View / metadata operators (no launch) are left out.
usage: python tools/aten_origins.py cfg3 [cfg4 ...]   (GPU box)"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

NO_LAUNCH = ("view", "reshape", "as_strided", "slice", "select", "detach", "alias", "t.default", "transpose", "expand", "unsqueeze",
             "squeeze", "permute", "_unsafe_view", "empty", "is_", "size", "stride", "numel", "dim", "split", "unbind", "narrow",
             "lift_fresh", "_local_scalar", "record_stream", "set_", "resize_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in NO_LAUNCH):
            fr = [f for f in traceback.extract_stack() if ("/kgcn_amd/" in f.filename or f.filename.endswith("bench.py"))]
            where = " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in fr[-3:][::-1])
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
            key = (name, where, str(shapes))
            self.seen[key] = self.seen.get(key, 0) + 1
        return func(*args, **(kwargs or {}))


for cfg in sys.argv[1:] or ["cfg3"]:
    args = bench.build_parser().parse_args(["--config", cfg, "--eager", "--no-cpu-baseline"])
    ctx = bench.Ctx(args)
    wl = {"cfg1": bench.Cfg1, "cfg3": bench.Cfg3, "cfg4": bench.Cfg4, "cfg5": bench.Cfg5}[cfg](args, ctx)
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        wl.step()
    torch.cuda.synchronize()
    print("==", cfg)
    for (name, where, shapes), n in log.seen.items():
        print("%dx %-28s %s   %s" % (n, name, shapes, where))
