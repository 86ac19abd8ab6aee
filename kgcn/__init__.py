"""Import-path alias of the reference package name for the hot path only.

The reference's model plugins and tools address the layer / op API as `kgcn.layers`,
`kgcn.bspmm_call`, `kgcn.bconv_call`, `kgcn.batched_call` (example_model/model.py:1-9 `import
kgcn.layers`, KNIME/GCN-K/py/gcn_infer.py:530-535 pokes `kgcn.layers.enabled_bspmm`).  These four
modules -- and nothing else of the reference package -- resolve here, to the MI355X implementation in
`kgcn_amd` (same objects, not copies: module flags set through either name are seen by both).
Everything else of the reference's `kgcn` package (CLI, trainer, preprocessing, visualisation
front ends) is out of scope and intentionally absent.
"""
from kgcn_amd import *  # noqa: F401,F403
from kgcn_amd import __doc__ as _impl_doc  # noqa: F401
