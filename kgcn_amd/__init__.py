"""kgcn_amd -- MI355X (gfx950) implementation of kGCN's batched graph-convolution hot path.

Scope: kgcn.layers GraphConv / GraphDense / GINAggregate / GraphGather, the op
wrappers bspmm_call / bconv_call / batched_call and their gradients, behind the reference's own
layer/op API.  The arithmetic runs in hand-written HIP kernels (kgcn_amd/csrc) reached through
the C ABI of include/kgcn_hip.h; importing the package fails if that library is not built.
"""
from . import _lib  # noqa: F401  -- loud failure when libkgcn_hip.so is missing
from . import layers, ops
from .batched_csr import BatchedAdjacency, BatchedCSR, PackedAdjacencyCache, as_batched_adjacency

__all__ = ["layers", "ops", "BatchedAdjacency", "BatchedCSR", "PackedAdjacencyCache", "as_batched_adjacency"]
__version__ = "0.1.0"
