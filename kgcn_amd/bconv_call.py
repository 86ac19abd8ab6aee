"""Op `Bconv` behind the reference's wrapper API (kgcn/bconv_call.py:6-23).

    BatchedConv().call(sp_matrices, dense_matrices, adjoint_a=False, adjoint_b=False)

sp_matrices[b][ch] / dense_matrices[b][ch] are batch-major nested lists (the reference flattens
them batch-major and passes dim_matrices=[numChannels, batchSize], :12-20).  One fused launch
computes out[b] = sum_ch S[b][ch] @ rhs[b][ch]; returns the list of batchSize dense [M, D]
tensors.  Gradient as registered by the reference (kgcn/bconv_call.py:30-70).
"""
import torch

from . import ops
from .batched_csr import BatchedAdjacency
from .bspmm_call import _diff_values


class BatchedConv:
    def __init__(self):
        from . import _lib  # noqa: F401

    def call(self, sp_matrices, dense_matrices, adjoint_a=False, adjoint_b=False):
        if adjoint_a or adjoint_b:
            raise NotImplementedError("Bconv is only used with adjoint_a=adjoint_b=False "
                                      "(kgcn/layers.py:77, 435)")
        B = len(dense_matrices)
        C = len(dense_matrices[0])
        # [B, K, C*D]: channel ch of graph b in columns ch*D..(ch+1)*D
        rows = [torch.cat(list(dense_matrices[b]), dim=1) for b in range(B)]
        rhs = torch.stack(rows)
        K, CD = rhs.shape[1], rhs.shape[2]
        D = CD // C
        values = None
        if isinstance(sp_matrices, BatchedAdjacency):
            adj = sp_matrices
        else:
            adj = BatchedAdjacency.from_adjs(sp_matrices, device=rhs.device)
            # differentiable .values (kgcn/bconv_call.py:55-67 registers d values for every graph-channel)
            per_ch = [_diff_values([sp_matrices[b][ch] for b in range(B)], adj.channels[ch]) for ch in range(C)]
            if any(v is not None for v in per_ch):
                values = per_ch
        out = ops.bconv(adj, rhs.reshape(B * K, CD).contiguous(), D, values)
        return list(out.reshape(B, adj.n_nodes, D).unbind(0))
