"""Why does autograd CLONE the BN gradients (new_empty_strided + copy_ per step in cfg3 / cfg4)?  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from kgcn_amd import layers, ops

dev = torch.device("cuda:0")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = []

    def __torch_dispatch__(self, func, types, a=(), kw=None):
        n = str(func)
        if n.startswith(("aten.copy_", "aten.new_empty_strided", "aten.clone", "aten.add")):
            self.seen.append((n, [tuple(t.shape) for t in a if torch.is_tensor(t)][:2]))
        return func(*a, **(kw or {}))


for defer in (False, True):
    for act in (None, "sigmoid"):
        bn = layers.GraphBatchNormalization(learning_phase=0, activation=act)
        x = torch.randn(64, 10, 50, device=dev, requires_grad=True)
        en = torch.randint(0, 11, (64,), device=dev)
        bn(x, enabled_node_nums=en)
        lin = torch.nn.Linear(50, 3).to(dev)
        for rep in range(2):
            for p in list(bn.parameters()) + list(lin.parameters()):
                p.grad = None
            x.grad = None
            y = bn(x, enabled_node_nums=en)
            cost = lin(y).sum()
            log = Log()
            with log:
                if defer:
                    with ops.deferred_reductions(root=cost):
                        cost.backward()
                else:
                    cost.backward()
            print("defer", defer, "act", act, "rep", rep, log.seen)
