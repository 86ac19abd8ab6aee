"""ctypes binding of libkgcn_hip.so (the C ABI declared in include/kgcn_hip.h).

There is no CPU fallback: if the shared library is missing or lacks a symbol, importing this
module raises.  PyTorch is used only for device memory and streams; every call passes raw device
pointers and the current HIP stream.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Development overrides.  KGCN_HIP_LIB: load another build of the library (tools/variant_bench.py times alternative builds).
# KGCN_DENSE_ROUTE / KGCN_GEMM3_MW: kernel-routing knobs that the library itself only honours when it was compiled with
# -DKGCN_DEV_KNOBS (make DEV_KNOBS=1) -- the shipped build ignores them.  Anything set here is reported by
# active_overrides() (bench.py prints it in its JSON line) and warned about once at import, because a stray variable
# changes summation order / which binary produced the numbers.
DEV_ENV_VARS = ("KGCN_HIP_LIB", "KGCN_DENSE_ROUTE", "KGCN_GEMM3_MW", "KGCN_GEMM3_CUT", "KGCN_WGRADX", "KGCN_WGRADN", "KGCN_GEMMH", "KGCN_WGRADL", "KGCN_SPMM_BLOCKS", "KGCN_GIN_JOIN", "KGCN_GIN_DOT")
LIB_PATH = os.environ.get("KGCN_HIP_LIB") or os.path.join(_HERE, "csrc", "libkgcn_hip.so")


def active_overrides():
    """Development environment switches that are set in this process (name -> value); {} in a clean environment."""
    return {k: os.environ[k] for k in DEV_ENV_VARS if os.environ.get(k)}


if active_overrides():
    import warnings
    warnings.warn("kgcn_amd: development overrides active: %r (KGCN_HIP_LIB swaps the native library; the routing knobs "
                  "act only on a -DKGCN_DEV_KNOBS build)" % (active_overrides(),), RuntimeWarning, stacklevel=2)

c_f32p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32


# KGCN_HIP_ABI_VERSION this binding was written against (include/kgcn_hip.h); the library must report the same
ABI_VERSION = 2


class CsrBatch(ctypes.Structure):
    """struct kgcn_csr_batch (include/kgcn_hip.h)."""
    _fields_ = [
        ("num_graphs", c_i32),
        ("rows", c_i32),
        ("cols", c_i32),
        ("max_nnz_per_graph", c_i32),
        ("row_pad", c_i32),
        ("reserved_", c_i32),
        ("nnz", c_i64),
        ("rowptr", ctypes.c_void_p),
        ("cv", ctypes.c_void_p),
        ("slots", ctypes.c_void_p),
        ("graph_ptr", ctypes.c_void_p),
        ("block_ptr", ctypes.c_void_p),
        ("num_blocks", c_i32),
        ("block_rows_max", c_i32),
    ]


_CSRP = ctypes.POINTER(CsrBatch)


class StackLayer(ctypes.Structure):
    """struct kgcn_stack_layer (include/kgcn_hip.h)."""
    _fields_ = [("kind", c_i32), ("act", c_i32), ("din", c_i32), ("dout", c_i32), ("w", ctypes.c_void_p),
                ("b", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("var", ctypes.c_void_p), ("eps", ctypes.c_float),
                ("route", c_i32)]


_STKP = ctypes.POINTER(StackLayer)

class AdamSegment(ctypes.Structure):
    """struct kgcn_adam_segment (include/kgcn_hip.h)."""
    _fields_ = [("grad", ctypes.c_void_p), ("offset", c_i64), ("numel", c_i64)]


class Copy2dJob(ctypes.Structure):
    """struct kgcn_copy2d_job (include/kgcn_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", c_i64), ("cols", c_i64), ("src_ld", c_i64), ("dst_ld", c_i64)]


class WtableJob(ctypes.Structure):
    """struct kgcn_wtable_job (include/kgcn_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("w_ld", c_i64), ("trans_w", c_i32), ("k", c_i32), ("n", c_i32), ("k_w", c_i32),
                ("table", ctypes.c_void_p), ("extra_row", ctypes.c_void_p)]


ASSEMBLE_MAX_CSR, ASSEMBLE_MAX_TABLES = 4, 6


class AssemblePlan(ctypes.Structure):
    """struct kgcn_assemble_plan (include/kgcn_hip.h)."""
    _fields_ = [("num_csr", c_i32), ("num_tables", c_i32),
                ("src", _CSRP * ASSEMBLE_MAX_CSR),
                ("dst_rowptr", ctypes.c_void_p * ASSEMBLE_MAX_CSR), ("dst_cv", ctypes.c_void_p * ASSEMBLE_MAX_CSR),
                ("dst_cv_capacity", c_i64 * ASSEMBLE_MAX_CSR), ("dst_slots", ctypes.c_void_p * ASSEMBLE_MAX_CSR),
                ("dst_graph_ptr", ctypes.c_void_p * ASSEMBLE_MAX_CSR),
                ("table", ctypes.c_void_p * ASSEMBLE_MAX_TABLES), ("table_out", ctypes.c_void_p * ASSEMBLE_MAX_TABLES),
                ("row_floats", c_i64 * ASSEMBLE_MAX_TABLES)]


_ASMP = ctypes.POINTER(AssemblePlan)
_PTRP = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list EVERY function include/kgcn_hip.h declares
# (tests/test_abi.py parses the header and compares).
SIGNATURES = {
    "kgcn_abi_version": (ctypes.c_int, []),
    "kgcn_csr_batch_size": (c_i64, []),
    "kgcn_last_error": (ctypes.c_char_p, []),
    "kgcn_build_arch": (ctypes.c_char_p, []),
    "kgcn_bspmm_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i64, c_i64, c_i32, c_f32p, c_i64, c_i64,
                                      ctypes.c_float, ctypes.c_void_p]),
    "kgcn_bconv_f32": (ctypes.c_int, [_CSRP, c_i32, c_f32p, c_i64, c_i64, c_i64, c_i32, c_f32p,
                                      c_i64, c_i64, ctypes.c_void_p]),
    "kgcn_spmm_values_grad_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i64, c_i64, c_f32p, c_i64,
                                                 c_i64, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_dense_fwd_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32,
                                          c_f32p, c_f32p, c_i32, c_i64, ctypes.c_void_p]),
    "kgcn_dense_wgrad_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32]),
    "kgcn_dense_bwd_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "kgcn_dense_bwd_dot_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i32, c_i64, c_f32p, c_i64, c_i64, c_i32, c_i32, c_f32p, c_i64, c_f32p, c_i64,
                                              c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_i64, c_i32, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_f32p, c_i32, c_i64, c_f32p, c_i64, c_i64, c_i32, c_i32,
                                          c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_f32p, ctypes.c_void_p, c_i64, c_i32,
                                          ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_wgrad_f32": (ctypes.c_int, [c_f32p, c_i64, c_f32p, c_i64, c_i64, c_i32, c_i32,
                                            c_f32p, c_f32p, ctypes.c_void_p, c_i64,
                                            ctypes.c_void_p]),
    "kgcn_graphconv_fused_supported": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    "kgcn_graphconv_fwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_f32p, c_f32p, c_i32, c_i32,
                                              c_f32p, ctypes.c_void_p]),
    "kgcn_graphconv_bwd_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "kgcn_graphconv_bwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_f32p, c_f32p, c_i32, c_i32,
                                              c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_i64,
                                              ctypes.c_void_p]),
    "kgcn_gin_aggregate_f32": (ctypes.c_int, [_CSRP, c_i32, c_f32p, c_i32, c_f32p, c_f32p,
                                              ctypes.c_void_p]),
    "kgcn_csr_gather_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_csr_gather_graphs": (ctypes.c_int, [_CSRP, c_i32p, c_i32, c_i32p, c_i32p, c_i64, c_i32p, c_i32p,
                                              ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_batch_assemble_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_batch_assemble": (ctypes.c_int, [_ASMP, c_i32p, c_i32, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_graph_maxpool_fwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i32, c_f32p, ctypes.c_float,
                                                  ctypes.c_void_p]),
    "kgcn_graph_maxpool_bwd_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "kgcn_graph_maxpool_bwd_f32": (ctypes.c_int, [_CSRP, _CSRP, c_f32p, c_f32p, c_i32, c_f32p,
                                                  ctypes.c_float, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_gat_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "kgcn_gat_fwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i32, c_f32p, c_f32p, ctypes.c_float, ctypes.c_void_p, c_i64,
                                        ctypes.c_void_p]),
    "kgcn_gat_bwd_f32": (ctypes.c_int, [_CSRP, _CSRP, c_f32p, c_i32, c_f32p, c_f32p, c_f32p, ctypes.c_float, c_f32p,
                                        ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_gram_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_gram_fwd_f32": (ctypes.c_int, [c_f32p, c_i32, c_i32, c_i32, c_f32p, c_f32p, ctypes.c_void_p]),
    "kgcn_gram_bwd_f32": (ctypes.c_int, [c_f32p, c_i32, c_i32, c_i32, c_f32p, c_f32p, c_f32p, ctypes.c_float, c_f32p,
                                         ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_graph_gather_fwd_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_f32p,
                                                 ctypes.c_void_p]),
    "kgcn_graph_gather_fwd_ld_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_f32p, c_i64, ctypes.c_void_p]),
    "kgcn_graph_gather_bwd_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_f32p,
                                                 ctypes.c_void_p]),
    "kgcn_bconv_fanout_f32": (ctypes.c_int, [_CSRP, c_i32, c_f32p, c_f32p, c_i64, c_i64, c_i32, c_i32, c_f32p, c_i64, c_i64, c_i64,
                                             ctypes.c_void_p]),
    "kgcn_bconv_act_f32": (ctypes.c_int, [_CSRP, c_i32, c_f32p, c_i64, c_i64, c_i64, c_i32, c_f32p, c_i64, c_i64, c_i32,
                                          ctypes.c_void_p]),
    "kgcn_bspmm_dact_f32": (ctypes.c_int, [_CSRP, c_f32p, c_f32p, c_i64, c_i64, c_i32, c_i32, c_f32p, c_i64, c_i64,
                                           ctypes.c_float, ctypes.c_void_p]),
    "kgcn_dense_fwd_act_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_f32p, c_i32,
                                              c_i64, c_i32, ctypes.c_void_p]),
    "kgcn_dense_fwd_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "kgcn_dense_fwd_ws_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_f32p, c_i32,
                                             c_i64, c_i32, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_dx_dact_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_i64, c_i32,
                                              c_f32p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_dx_dact_tab_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_i64,
                                                  c_i32, c_f32p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_dx_dact_gather_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "kgcn_dense_dx_dact_gather_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32,
                                                     c_f32p, c_i64, c_i32, c_f32p, ctypes.c_void_p, c_i64, c_i32, ctypes.c_void_p]),
    "kgcn_dense_dx_dact_dot_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "kgcn_dense_dx_dact_dot_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "kgcn_dense_dx_dact_dot_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_i64, c_i32,
                                                  c_f32p, ctypes.c_void_p, c_i64, c_i32, c_f32p, ctypes.c_void_p, c_i64,
                                                  ctypes.c_void_p]),
    "kgcn_dense_fwd_tab_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, c_i32, c_f32p, c_f32p, c_i32,
                                              c_i64, c_i32, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_wtable_split_multi": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_void_p]),
    "kgcn_act_fwd_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_act_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_graph_bn_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_graph_bn_stats_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_i32p, c_f32p, c_f32p, ctypes.c_void_p, c_i64,
                                               ctypes.c_void_p]),
    "kgcn_graph_bn_apply_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                               ctypes.c_float, c_f32p, ctypes.c_void_p]),
    "kgcn_graph_bn_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_i32p, c_f32p, c_f32p, c_f32p,
                                             ctypes.c_float, c_i32, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_i64,
                                             ctypes.c_void_p]),
    "kgcn_coo_pack_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i32]),
    "kgcn_coo_pack_f32": (ctypes.c_int, [c_i32p, c_i32p, c_i32p, c_f32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32p,
                                         ctypes.c_void_p, c_i32p, c_i32p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_csr_pad4_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "kgcn_csr_pad4": (ctypes.c_int, [_CSRP, c_i32p, ctypes.c_void_p, c_i64, c_i32p, c_i32p, c_i32p, ctypes.c_void_p,
                                     c_i64, ctypes.c_void_p]),
    "kgcn_graph_bn_apply_act_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i32, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                   ctypes.c_float, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_graph_bn_bwd_dact_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32, c_i64, c_i32, c_i32, c_i32p, c_f32p, c_f32p,
                                                  c_f32p, ctypes.c_float, c_i32, c_f32p, c_f32p, c_f32p, ctypes.c_void_p,
                                                  c_i64, ctypes.c_void_p]),
    "kgcn_ragged_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_ragged_plan": (ctypes.c_int, [_CSRP, c_i32p, c_i32p, c_i32, c_i32p, c_i32p, ctypes.c_void_p, c_i64,
                                        ctypes.c_void_p]),
    "kgcn_ragged_compact_csr": (ctypes.c_int, [_CSRP, c_i32p, c_i32, c_i32p, c_i32p, c_i32, c_i32p, c_i32p, c_i64, c_i32p,
                                               ctypes.c_void_p]),
    "kgcn_ragged_compact_csr_pair": (ctypes.c_int, [_CSRP, _CSRP, c_i32p, c_i32, c_i32p, c_i32p, c_i32, c_i32p, c_i32p, c_i64, c_i32p,
                                                    c_i32p, c_i64, c_i32p, ctypes.c_void_p]),
    "kgcn_ragged_block_rows": (c_i32, []),
    "kgcn_ragged_num_blocks": (c_i32, [c_i32]),
    "kgcn_ragged_blocks": (ctypes.c_int, [c_i32p, c_i32, c_i32, c_i32p, ctypes.c_void_p]),
    "kgcn_ragged_compact_rows_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32, c_i32, c_i32, c_i32p, c_i32, c_f32p,
                                                    ctypes.c_void_p]),
    "kgcn_ragged_compact_rows_aug_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32, c_i32, c_i32, c_i32p, c_i32, c_f32p, c_i32,
                                                        ctypes.c_void_p]),
    "kgcn_ragged_expand_rows_f32": (ctypes.c_int, [c_f32p, c_i32, c_i32, c_i32, c_i32p, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_ragged_gather_fwd_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i64, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_ragged_gather_bwd_workspace_bytes": (c_i64, [c_i32]),
    "kgcn_ragged_gather_bwd_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32p,
                                                  ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_mfma_products": (ctypes.c_int, [c_i32, c_i64, c_i32, c_i32]),
    "kgcn_copy2d_multi_f32": (ctypes.c_int, [ctypes.c_void_p, c_i32, ctypes.c_void_p]),
    "kgcn_hbm_probe": (ctypes.c_int, [c_i32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_reduce_defer": (ctypes.c_int, [c_i32]),
    "kgcn_reduce_pending": (ctypes.c_int, []),
    "kgcn_reduce_flush": (ctypes.c_int, [ctypes.c_void_p]),
    "kgcn_loss_workspace_bytes": (c_i64, [c_i64]),
    "kgcn_masked_sigmoid_ce_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_i32, ctypes.c_float, c_f32p,
                                                  c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_masked_softmax_ce_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_i32, c_f32p, c_f32p, c_f32p,
                                                  ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_sparse_softmax_ce_f32": (ctypes.c_int, [c_f32p, ctypes.c_void_p, c_f32p, c_i64, c_i32, c_f32p, c_f32p, c_f32p,
                                                  ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_adam_tf_multi_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i64, ctypes.c_void_p, c_i32, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kgcn_loss_grad_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_f32p, ctypes.c_void_p]),
    "kgcn_adam_tf_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kgcn_augment_ones_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, ctypes.c_void_p]),
    "kgcn_augment_ones_bwd_f32": (ctypes.c_int, [c_f32p, c_i64, c_i32, c_i64, c_f32p, c_i64, ctypes.c_void_p]),
    "kgcn_gcn_stack_supported": (ctypes.c_int, [c_i32, c_i32, _STKP, c_i32]),
    "kgcn_gcn_stack_param_floats": (c_i64, [_STKP, c_i32]),
    "kgcn_gcn_stack_fwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i32p, _STKP, c_i32, _PTRP, c_f32p, ctypes.c_void_p]),
    "kgcn_gcn_stack_bwd_workspace_bytes": (c_i64, [c_i32, _STKP, c_i32]),
    "kgcn_gcn_stack_bwd_f32": (ctypes.c_int, [_CSRP, c_f32p, c_i32p, _STKP, c_i32, _PTRP, c_f32p, c_i32, c_f32p, c_f32p,
                                              ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_dense_wgrad_dact_supported": (ctypes.c_int, [c_i32, c_i32]),
    "kgcn_dense_wgrad_dact_f32": (ctypes.c_int, [c_f32p, c_i64, c_f32p, c_f32p, c_i64, c_i32, c_i64, c_i32, c_i32, c_f32p,
                                                 c_f32p, ctypes.c_void_p, c_i64, ctypes.c_void_p]),
    "kgcn_gin_aggregate_bwd_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "kgcn_gin_aggregate_bwd_f32": (ctypes.c_int, [_CSRP, c_i32, c_f32p, c_i32, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p,
                                                  c_i64, ctypes.c_void_p]),
    "kgcn_graph_gather_bwd_add_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_i32, c_i32, c_f32p, ctypes.c_void_p]),
    "kgcn_dot_workspace_bytes": (c_i64, [c_i64]),
    "kgcn_dot_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i64, c_f32p, ctypes.c_void_p, c_i64,
                                    ctypes.c_void_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "kgcn_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C kgcn_amd/csrc` (hipcc --offload-arch=gfx950). There is no "
            "CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("kgcn_amd: %s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.kgcn_abi_version() != ABI_VERSION:
        raise ImportError("kgcn_amd: ABI version mismatch: library %d, binding %d" % (lib.kgcn_abi_version(), ABI_VERSION))
    if lib.kgcn_csr_batch_size() != ctypes.sizeof(CsrBatch):
        raise ImportError("kgcn_amd: kgcn_csr_batch is %d bytes in the library, %d in the binding (stale build?)"
                          % (lib.kgcn_csr_batch_size(), ctypes.sizeof(CsrBatch)))
    return lib


lib = _load()


class KgcnHipError(RuntimeError):
    pass


def check(rc, what=""):
    """Raise on a non-zero status, with the library's message (error convention of the ABI)."""
    if rc != 0:
        msg = lib.kgcn_last_error()
        raise KgcnHipError("%s failed: %s" % (what or "kgcn_hip call",
                                              msg.decode() if msg else "unknown error"))


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def require_gpu(t, name):
    if not t.is_cuda:
        raise KgcnHipError(
            "%s must live on the GPU (cuda/HIP device); kgcn_amd has no CPU path" % name)
