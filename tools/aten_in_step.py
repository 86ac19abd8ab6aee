#!/usr/bin/env python3
"""Which torch (aten) operators run inside a training step, and from which line of kgcn_amd / bench.py: one step of a bench.py
configuration under a TorchDispatchMode that records the Python stack of every device operator (operators issued by the autograd
engine itself -- gradient seeds, materialised zero gradients, cloned gradients -- show "<autograd engine>").
usage: python tools/aten_in_step.py --config cfg3 [--graphed] [--who] [bench args]
  --graphed  the step as GraphedTrainStep captures it (its _eager body, run once more outside a capture) instead of bench.py's eager step
  --who      which parameter's AccumulateGrad made a copy (note: the hooks this registers change AccumulateGrad's stealing rule)
tests/test_gpu_step_no_aten.py runs log_step() on the captured steps of cfg1 / cfg3 / cfg4 / cfg5 and fails on any entry."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# operators that launch nothing: views, metadata, allocations
SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.t.", "aten.select", "aten.slice",
        "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.empty", "aten.transpose", "aten._local_scalar",
        "aten.narrow", "aten.permute", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.split", "aten.unbind", "aten.record_stream",
        "aten.new_empty.", "aten.new_empty_strided.", "aten.empty_like", "aten.empty_strided", "aten.unflatten", "aten.flatten",
        "aten.chunk", "aten.stride", "aten.size", "aten.numel", "aten.dim", "aten.storage_offset", "aten.is_contiguous")


def log_step(step_fn, on_copy=None):
    """-> OrderedDict {(operator, where, shapes): count} of the device operators step_fn() issues (SKIP excluded)."""
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = collections.OrderedDict()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, a=(), kw=None):
            name = str(func)
            if not name.startswith(SKIP):
                st = [f for f in traceback.extract_stack() if ("kgcn_amd/" in f.filename or f.filename.endswith("bench.py"))]
                where = "%s:%d %s" % (os.path.relpath(st[-1].filename, ROOT), st[-1].lineno, st[-1].name) if st else "<autograd engine>"
                shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]
                key = (name, where, str(shapes))
                seen[key] = seen.get(key, 0) + 1
                if on_copy is not None and name.startswith("aten.copy_"):
                    on_copy(shapes)
            return func(*a, **(kw or {}))

    with Log():
        step_fn()
    torch.cuda.synchronize()
    return seen


def build(config, extra=(), graphed=True):
    """-> (workload, step function) of a bench.py configuration; graphed: the body GraphedTrainStep captured."""
    import torch
    import bench
    args = bench.build_parser().parse_args(["--config", config] + list(extra) + ([] if graphed else ["--eager"]) + ["--no-cpu-baseline"])
    ctx = bench.Ctx(args)
    wl = {"cfg1": bench.Cfg1, "cfg3": bench.Cfg3, "cfg4": bench.Cfg4, "cfg5": bench.Cfg5}[config](args, ctx)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    if not graphed:
        return wl, wl.step
    assert wl.graph_step is not None, "this configuration has no captured step"
    return wl, wl.graph_step._eager


def main():
    argv = [a for a in sys.argv[1:] if a not in ("--graphed", "--who")]
    config = argv[argv.index("--config") + 1] if "--config" in argv else "cfg4"
    extra = [a for i, a in enumerate(argv) if a != "--config" and (i == 0 or argv[i - 1] != "--config")]
    wl, step = build(config, extra, graphed="--graphed" in sys.argv)
    fired = []
    if "--who" in sys.argv:
        model = wl.model.model if hasattr(wl.model, "model") else wl.model
        for n_, p_ in model.named_parameters():
            p_.register_post_accumulate_grad_hook(lambda t, n_=n_: fired.append((n_, tuple(t.shape))))
    seen = log_step(step, on_copy=(lambda shapes: fired.append(("<copy_>", shapes))) if "--who" in sys.argv else None)
    for (name, where, shapes), n in seen.items():
        print("%2d x %-34s %-40s <- %s" % (n, name, shapes, where))
    if not seen:
        print("no torch operator launches anything inside the step")
    if fired:
        print("order of parameter accumulations and copies:", fired)


if __name__ == "__main__":
    main()
