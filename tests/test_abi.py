"""CPU-side checks of the drop-in boundary: libkgcn_hip.so loads (no GPU needed), exports every
symbol include/kgcn_hip.h declares, the Python binding lists exactly those symbols, shape queries
answer, and argument validation fails loudly (no compute launches here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kgcn_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(kgcn_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def header_abi_version():
    return int(re.search(r"#define\s+KGCN_HIP_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))


def consumer_binary():
    """tests/abi_consumer.bin, (re)built with plain gcc when it is missing or older than its source / the header."""
    import __graft_entry__ as g
    out = os.path.join(ROOT, "tests", "abi_consumer.bin")
    srcs = (os.path.join(ROOT, "tests", "abi_consumer.c"), HEADER)
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        if not os.path.exists("/usr/bin/gcc"):
            pytest.skip("no gcc and no prebuilt abi_consumer.bin")
        g.build_abi_consumer()
    return out


def run_consumer(mode):
    import subprocess
    p = subprocess.run([consumer_binary(), mode], capture_output=True, text=True, timeout=120)
    facts = {}
    for line in p.stdout.splitlines():
        k, _, v = line.partition(" ")
        facts[k] = v
    return p.returncode, facts, p.stdout + p.stderr


def test_c_consumer_sees_the_layout_the_binding_mirrors():
    """A plain-C program compiled against include/kgcn_hip.h: its sizeof / offsetof of kgcn_csr_batch, the library's
    kgcn_csr_batch_size() and the ctypes mirror in kgcn_amd/_lib.py must all agree -- layout drift that a ctypes-only test
    cannot see (the struct grew in ABI version 2)."""
    from kgcn_amd import _lib
    rc, facts, text = run_consumer("layout")
    assert rc == 0 and "OK" in facts, text
    assert int(facts["header_abi_version"]) == int(facts["library_abi_version"]) == _lib.ABI_VERSION
    assert int(facts["sizeof_csr_batch"]) == int(facts["library_csr_batch_size"]) == ctypes.sizeof(_lib.CsrBatch)
    for field in ("nnz", "rowptr", "cv", "slots", "graph_ptr", "block_ptr", "num_blocks", "block_rows_max"):
        assert int(facts["offsetof_" + field]) == getattr(_lib.CsrBatch, field).offset, field
    assert int(facts["sizeof_wtable_job"]) == ctypes.sizeof(_lib.WtableJob)
    assert int(facts["offsetof_wtable_job_table"]) == _lib.WtableJob.table.offset
    assert int(facts["offsetof_wtable_job_extra_row"]) == _lib.WtableJob.extra_row.offset
    assert "NULL" in facts["last_error"]


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("kgcn_bspmm_f32", "kgcn_bconv_f32", "kgcn_spmm_values_grad_f32", "kgcn_dense_fwd_f32",
                 "kgcn_dense_wgrad_f32", "kgcn_graphconv_fwd_f32", "kgcn_graphconv_bwd_f32",
                 "kgcn_gin_aggregate_f32", "kgcn_graph_gather_fwd_f32", "kgcn_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from kgcn_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "libkgcn_hip.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared_functions(), "binding and header disagree"
    assert _lib.lib.kgcn_abi_version() == _lib.ABI_VERSION == header_abi_version()
    assert _lib.lib.kgcn_csr_batch_size() == ctypes.sizeof(_lib.CsrBatch)
    assert _lib.lib.kgcn_build_arch() == b"gfx950"


def test_code_object_is_gfx950_only():
    from kgcn_amd import _lib
    import re
    blob = open(_lib.LIB_PATH, "rb").read()
    # the offload bundle lists one entry id per device code object: "hipv4-amdgcn-amd-amdhsa--<arch>" (bare arch names also
    # occur as strings of rocPRIM's host-side tuning tables -- those are not code)
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets
    for other in (b"sm_90", b"nvptx"):
        assert other not in blob


def test_shape_queries_and_validation():
    from kgcn_amd import _lib
    lib = _lib.lib
    assert lib.kgcn_graphconv_fused_supported(32, 64, 64, 100) == 1
    assert lib.kgcn_graphconv_fused_supported(10, 4, 52, 24) == 1
    assert lib.kgcn_graphconv_fused_supported(10, 3, 50, 24) == 1       # any width <= 64 (scalar tile path)
    assert lib.kgcn_graphconv_fused_supported(10, 65, 50, 24) == 0
    assert lib.kgcn_graphconv_fused_supported(50, 64, 64, 160) == 0      # N > 32
    assert lib.kgcn_graphconv_fused_supported(32, 128, 64, 100) == 0
    assert lib.kgcn_dense_wgrad_workspace_bytes(3_200_000, 64, 64) > 0
    assert lib.kgcn_graphconv_bwd_workspace_bytes(100_000, 64, 64) >= 256 * (64 * 64 + 64) * 4   # one partial per CU
    assert lib.kgcn_dot_workspace_bytes(10) > 0
    # validation happens before any launch: NULL descriptor / bad sizes -> status + message
    rc = lib.kgcn_bspmm_f32(None, None, 0, 0, 4, None, 0, 0, 0.0, None)
    assert rc != 0 and b"NULL" in lib.kgcn_last_error()
    d = _lib.CsrBatch(-1, 4, 4, 0, 0, 0, 0, None, None, None, None)
    assert lib.kgcn_bspmm_f32(ctypes.byref(d), None, 4, 16, 4, None, 4, 16, 0.0, None) != 0
    d = _lib.CsrBatch(2, 4, 4, 0, 0, 0, 0, 1, None, None, None)                            # fake non-NULL rowptr
    assert lib.kgcn_bspmm_f32(ctypes.byref(d), None, 4, 16, 4, None, 4, 16, 0.0, None) != 0
    assert b"NULL" in lib.kgcn_last_error()
    assert lib.kgcn_bspmm_f32(ctypes.byref(d), 16, 2, 16, 4, 16, 4, 16, 0.0, None) != 0   # ld < d
    assert lib.kgcn_bspmm_f32(ctypes.byref(d), 16, 4, 16, 4, 16, 4, 16, 0.5, None) != 0   # beta
    pd = _lib.CsrBatch(2, 4, 4, 16, 4, 0, 32, 1, 1, 1, 1)                        # row-padded batch
    assert lib.kgcn_bspmm_f32(ctypes.byref(pd), 16, 4, 16, 4, 16, 4, 16, 0.0, None) != 0
    assert b"row_pad" in lib.kgcn_last_error()                            # plain kernels refuse it
    with pytest.raises(_lib.KgcnHipError):
        _lib.check(lib.kgcn_dense_fwd_f32(None, 10, 0, 0, None, 0, 0, None, None, 4, 4, None), "dense")
    rc = lib.kgcn_graphconv_fwd_f32(ctypes.byref(_lib.CsrBatch(1, 50, 50, 12, 4, 0, 0, 1, None, 1, 1)), None, None,
                                    None, 64, 64, None, None)
    assert rc != 0 and b"not supported" in lib.kgcn_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    from kgcn_amd import _lib, layers
    layer = layers.GraphConv(8, 1)
    adjs = [[([[0, 0]], [1.0], [4, 4])] for _ in range(2)]
    with pytest.raises((_lib.KgcnHipError, RuntimeError, AssertionError)):
        layer(torch.zeros((2, 4, 4)), adj=adjs)                          # CPU tensors are refused
    # nothing under kgcn_amd/ imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kgcn_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f
