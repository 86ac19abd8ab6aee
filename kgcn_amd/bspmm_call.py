"""Op `Bspmm` behind the reference's wrapper API (kgcn/bspmm_call.py:6-16).

    BatchedSpMM().call(sp_matrices, dense_matrices, adjoint_a=False, adjoint_b=False)

sp_matrices: T sparse matrices (objects with .indices [nnz,2] / .values [nnz] / .dense_shape [2],
or (indices, values, dense_shape) tuples); dense_matrices: T dense [K, D] tensors (or one
[T, K, D] tensor).  Returns a list of T dense [M, D] tensors, like the TF op.  The gradient the
reference registers for the op (kgcn/bspmm_call.py:22-57: d rhs via Bspmm(adjoint_a=True),
d values via gather-multiply-reduce) is provided by kgcn_amd.ops._SpMM.
"""
import torch

from . import ops
from .batched_csr import BatchedCSR, _as_triple


def _stack_dense(dense_matrices, adjoint_b):
    if torch.is_tensor(dense_matrices):
        d = dense_matrices
    else:
        d = torch.stack(list(dense_matrices))
    return d.transpose(1, 2) if adjoint_b else d


def _diff_values(sp_matrices, csr):
    """If any .values is a tensor that requires grad, return them concatenated in CSR order."""
    vals = [_as_triple(m)[1] for m in sp_matrices]
    if not any(torch.is_tensor(v) and v.requires_grad for v in vals):
        return None
    cat = torch.cat([torch.as_tensor(v, dtype=torch.float32, device=csr.device).reshape(-1)
                     for v in vals])
    return cat if csr.perm is None else cat[csr.perm]


class BatchedSpMM:
    def __init__(self):
        from . import _lib  # noqa: F401  (fails loudly if libkgcn_hip.so is missing)

    def call(self, sp_matrices, dense_matrices, adjoint_a=False, adjoint_b=False):
        dense = _stack_dense(dense_matrices, adjoint_b).contiguous()
        csr = sp_matrices if isinstance(sp_matrices, BatchedCSR) else \
            BatchedCSR.from_coo_list(list(sp_matrices), device=dense.device)
        values = None if isinstance(sp_matrices, BatchedCSR) else _diff_values(sp_matrices, csr)
        if adjoint_a:
            if values is not None:
                csr = csr.with_values(values)       # keeps gradients to rhs only in this form
                values = None
            csr = csr.transpose()
        out = ops.bspmm(csr, dense, values)
        return list(out.unbind(0))
