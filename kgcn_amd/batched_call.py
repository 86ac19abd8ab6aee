"""Ops of the reference's `batched.so` behind its wrapper API (kgcn/batched_call.py:6-27):
BatchedSpMM (same contract as kgcn_amd.bspmm_call) and BatchedSpMDT, whose dense operand is ONE
stacked [T*K, D] tensor (kgcn/layers.py:99-102).  Returns a list of T dense [M, D] tensors
(kgcn/batched_call.py:70 indexes the gradients per graph).  Gradient: stacked Bspmm(adjoint_a=True)
(kgcn/batched_call.py:60-64).
"""
from . import ops
from .batched_csr import BatchedCSR
from .bspmm_call import BatchedSpMM, _diff_values  # noqa: F401  (same op contract)


class BatchedSpMDT:
    def __init__(self):
        from . import _lib  # noqa: F401

    def call(self, sp_matrices, dense_matrices, adjoint_a=False, adjoint_b=False):
        if adjoint_b:
            raise NotImplementedError("Bspmdt is only used with adjoint_b=False (kgcn/layers.py:102)")
        csr = sp_matrices if isinstance(sp_matrices, BatchedCSR) else \
            BatchedCSR.from_coo_list(list(sp_matrices), device=dense_matrices.device)
        values = None if isinstance(sp_matrices, BatchedCSR) else _diff_values(sp_matrices, csr)
        if adjoint_a:
            if values is not None:
                csr = csr.with_values(values)
                values = None
            csr = csr.transpose()
        out = ops.bspmm(csr, dense_matrices.contiguous(), values)      # [T*M, D]
        return list(out.reshape(csr.num_graphs, csr.rows, -1).unbind(0))
