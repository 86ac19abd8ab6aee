#!/usr/bin/env python3
"""Which torch (aten) operators run inside a training step, and from which line of kgcn_amd / bench.py: one eager step of a bench.py
configuration under a TorchDispatchMode that records the Python stack of every device operator (operators issued by the autograd
engine itself -- gradient seeds, materialised zero gradients, accumulation -- show "<autograd engine>").
usage: python tools/aten_in_step.py --config cfg3 [bench args]"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

GRAPHED = "--graphed" in sys.argv        # the step as GraphedTrainStep captures it (its _eager body, run once more outside a capture)
argv = [a for a in sys.argv[1:] if a != "--graphed"]
args = bench.build_parser().parse_args(argv + ([] if GRAPHED else ["--eager"]) + ["--no-cpu-baseline"])
ctx = bench.Ctx(args)
wl = {"cfg1": bench.Cfg1, "cfg3": bench.Cfg3, "cfg4": bench.Cfg4, "cfg5": bench.Cfg5}[args.config](args, ctx)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
step = wl.step
if GRAPHED:
    gs = getattr(wl, "graph_step", None) or getattr(wl, "gstep", None)
    assert gs is not None, "this configuration has no captured step"
    step = gs._eager
seen = collections.OrderedDict()
SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.t.", "aten.select", "aten.slice",
        "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.empty", "aten.transpose", "aten._local_scalar",
        "aten.narrow", "aten.permute", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.split", "aten.unbind", "aten.record_stream")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = [f for f in traceback.extract_stack() if ("kgcn_amd/" in f.filename or f.filename.endswith("bench.py"))]
            where = "%s:%d %s" % (os.path.relpath(st[-1].filename, ROOT), st[-1].lineno, st[-1].name) if st else "<autograd engine>"
            shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]
            key = (name, where, str(shapes))
            seen[key] = seen.get(key, 0) + 1
        return func(*a, **(kw or {}))


fired = []
if "--who" in sys.argv or os.environ.get("ATEN_WHO"):
    # which parameter's AccumulateGrad made a copy: the copy_ happens just before that parameter's post-accumulate hook fires
    model = wl.model.model if hasattr(wl.model, "model") else wl.model
    for n_, p_ in model.named_parameters():
        p_.register_post_accumulate_grad_hook(lambda t, n_=n_: fired.append((n_, tuple(t.shape))))
    _orig = Log.__torch_dispatch__

    def _td(self, func, types, a=(), kw=None):
        if str(func).startswith("aten.copy_"):
            fired.append(("<copy_>", [tuple(t.shape) for t in a if torch.is_tensor(t)][:1]))
        return _orig(self, func, types, a, kw)
    Log.__torch_dispatch__ = _td
with Log():
    step()
if fired:
    print("order of parameter accumulations and copies:", fired)
torch.cuda.synchronize()
for (name, where, shapes), n in seen.items():
    print("%2d x %-34s %-40s <- %s" % (n, name, shapes, where))
