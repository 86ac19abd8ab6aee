"""Raw C-ABI checks of the step-level entry points added in round 3: argument validation before any launch, chunking over the
per-launch limits (more than 16 weight tables, more than 32 Adam segments), empty inputs."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_wtable_split_multi_more_jobs_than_one_launch_takes():
    """20 operands (> KGCN_WTABLE_MAX_JOBS = 16: two launches) -- every table bit-identical to the single-operand split that
    kgcn_dense_fwd_ws_f32 performs, observed through the GEMM result of kgcn_dense_fwd_tab_f32."""
    from kgcn_amd import _lib
    from kgcn_amd._lib import lib, ptr, check, current_stream
    torch.manual_seed(0)
    m, d = 1500, 256
    x = torch.randn(m, d, device=dev())
    ws_list, tabs, jobs = [], [], (_lib.WtableJob * 20)()
    tb = int(lib.kgcn_dense_fwd_workspace_bytes(d, d))
    assert tb > 0
    for i in range(20):
        w = torch.randn(d, d, device=dev()) * 0.05
        t = torch.empty(tb // 4, device=dev())
        ws_list.append(w); tabs.append(t)
        jobs[i] = _lib.WtableJob(w.data_ptr(), d, i % 2, d, d, 0, t.data_ptr())
    check(lib.kgcn_wtable_split_multi(ctypes.cast(jobs, ctypes.c_void_p), 20, current_stream()), "kgcn_wtable_split_multi")
    for i in (0, 7, 16, 19):
        trans = i % 2
        y_tab, y_ws = torch.empty(m, d, device=dev()), torch.empty(m, d, device=dev())
        check(lib.kgcn_dense_fwd_tab_f32(ptr(x), m, d, d, ptr(ws_list[i]), d, trans, None, ptr(y_tab), d, d, 0, ptr(tabs[i]), tb,
                                         current_stream()), "tab")
        scratch = torch.empty(tb // 4, device=dev())
        check(lib.kgcn_dense_fwd_ws_f32(ptr(x), m, d, d, ptr(ws_list[i]), d, trans, None, ptr(y_ws), d, d, 0, ptr(scratch), tb,
                                        current_stream()), "ws")
        assert torch.equal(y_tab, y_ws)
    assert lib.kgcn_wtable_split_multi(ctypes.cast(jobs, ctypes.c_void_p), -1, current_stream()) != 0
    jobs[3].table = 0
    assert lib.kgcn_wtable_split_multi(ctypes.cast(jobs, ctypes.c_void_p), 20, current_stream()) != 0
    assert b"job 3" in lib.kgcn_last_error()


def test_adam_multi_more_segments_than_one_launch_takes():
    """40 parameter segments (> KGCN_ADAM_MAX_SEGMENTS = 32: two launches, ONE counter tick) against the packed-buffer update:
    identical parameters, moments and step count; floats between the segments stay untouched."""
    from kgcn_amd import _lib
    from kgcn_amd._lib import lib, ptr, check, current_stream
    rng = np.random.default_rng(1)
    sizes = rng.integers(1, 3000, 40)
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += -(-int(n) // 64) * 64
    total = off
    p0 = torch.from_numpy(rng.standard_normal(total).astype(np.float32)).to(dev())
    grads = [torch.from_numpy(rng.standard_normal(int(n)).astype(np.float32)).to(dev()) for n in sizes]
    flat_g = torch.zeros(total, device=dev())
    for g, o in zip(grads, offs):
        flat_g[o:o + g.numel()] = g
    res = []
    for multi in (False, True):
        p, m_, v_ = p0.clone(), torch.zeros(total, device=dev()), torch.zeros(total, device=dev())
        t = torch.zeros(1, dtype=torch.int64, device=dev())
        for step in range(3):
            if multi:
                segs = (_lib.AdamSegment * 40)()
                for i, (g, o) in enumerate(zip(grads, offs)):
                    segs[i] = _lib.AdamSegment(g.data_ptr(), o, g.numel())
                check(lib.kgcn_adam_tf_multi_f32(ptr(p), ptr(m_), ptr(v_), total, ctypes.cast(segs, ctypes.c_void_p), 40, 0.01, 0.9,
                                                 0.999, 1e-8, ptr(t), current_stream()), "multi")
            else:
                check(lib.kgcn_adam_tf_f32(ptr(p), ptr(flat_g), ptr(m_), ptr(v_), total, 0.01, 0.9, 0.999, 1e-8, ptr(t),
                                           current_stream()), "flat")
        res.append((p, m_, v_, int(t)))
    assert res[0][3] == res[1][3] == 3
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    gap = torch.ones(total, dtype=torch.bool)
    for n, o in zip(sizes, offs):
        gap[o:o + int(n)] = False
    assert torch.equal(res[1][0][gap.to(dev())], p0[gap.to(dev())])
    bad = (_lib.AdamSegment * 1)(_lib.AdamSegment(grads[0].data_ptr(), total - 1, 5))
    t = torch.zeros(1, dtype=torch.int64, device=dev())
    assert lib.kgcn_adam_tf_multi_f32(ptr(p0), ptr(p0), ptr(p0), total, ctypes.cast(bad, ctypes.c_void_p), 1, 0.01, 0.9, 0.999, 1e-8,
                                      ptr(t), current_stream()) != 0


def test_step_entry_points_validate_before_launching():
    from kgcn_amd import _lib
    from kgcn_amd._lib import lib, ptr, current_stream
    z = torch.zeros(64, device=dev())
    # gathered dX: rows must be whole graphs, the shape must have the fused form, pooled gradient required
    assert lib.kgcn_dense_dx_dact_gather_supported(4096, 256, 256) == 1
    assert lib.kgcn_dense_dx_dact_gather_supported(4096, 50, 50) == 0 and lib.kgcn_dense_dx_dact_gather_supported(100, 256, 256) == 0
    big = torch.zeros(4096 * 256, device=dev())
    tab = torch.zeros(int(lib.kgcn_dense_fwd_workspace_bytes(256, 256)) // 4, device=dev())
    args = dict(m=4096, dout=256, ld=256, w_ld=256, din=256, dx_ld=256)
    def call(grad, gp, n_nodes, act=2, m=4096):
        return lib.kgcn_dense_dx_dact_gather_f32(ptr(grad), ptr(gp), 256, n_nodes, ptr(big), m, 256, 256, ptr(big), 256, 256, ptr(big), 256,
                                                 act, ptr(big.clone()), ptr(tab), tab.numel() * 4, 0, current_stream())
    assert call(None, None, 10) != 0 and b"pooled gradient" in lib.kgcn_last_error()
    assert call(None, big, 7) != 0                                     # 4096 rows are not whole graphs of 7 nodes
    assert call(None, big, 16, act=0) != 0 and b"activation" in lib.kgcn_last_error()
    assert lib.kgcn_loss_grad_f32(ptr(z), None, None, 0, 64, ptr(z), current_stream()) != 0
    plan = _lib.AssemblePlan()
    plan.num_csr = 5
    assert lib.kgcn_batch_assemble(plan, None, 4, None, 0, current_stream()) != 0 and b"containers" in lib.kgcn_last_error()
    plan.num_csr = 0
    plan.num_tables = 1
    plan.row_floats[0] = 8
    assert lib.kgcn_batch_assemble(plan, ptr(torch.zeros(4, dtype=torch.int32, device=dev())), 4, None, 0, current_stream()) != 0
    torch.cuda.synchronize()


def test_dx_dact_dot_and_strided_gather_argument_checks():
    """kgcn_dense_dx_dact_dot_f32 / kgcn_graph_gather_fwd_ld_f32 / kgcn_ragged_blocks refuse what they cannot do, with a message."""
    from kgcn_amd._lib import lib, ptr, current_stream
    m, d = 16384, 256
    assert lib.kgcn_dense_dx_dact_dot_supported(m, d, d) == 1
    assert lib.kgcn_dense_dx_dact_dot_supported(m - 64, d, d) == 0 and lib.kgcn_dense_dx_dact_dot_supported(m, 50, 50) == 0
    big = torch.zeros(m * d, device=dev())
    tab = torch.zeros(int(lib.kgcn_dense_fwd_workspace_bytes(d, d)) // 4, device=dev())
    ws = torch.zeros(max(int(lib.kgcn_dense_dx_dact_dot_workspace_bytes(m, d)), 4) // 4, device=dev())
    out = torch.zeros(1, device=dev())

    def call(act=2, mm=m, dotx=big, wsb=None, dotx_ld=d):
        return lib.kgcn_dense_dx_dact_dot_f32(ptr(big), ptr(big), mm, d, d, ptr(big), d, d, ptr(dotx), dotx_ld, act, ptr(big.clone()),
                                              ptr(tab), tab.numel() * 4, 0, ptr(out), ptr(ws), ws.numel() * 4 if wsb is None else wsb,
                                              current_stream())
    assert call() == 0
    assert call(act=0) != 0 and b"activation" in lib.kgcn_last_error()
    assert call(mm=1000) != 0 and b"no fused form" in lib.kgcn_last_error()
    assert call(dotx=None) != 0 and b"NULL" in lib.kgcn_last_error()
    assert call(wsb=4) != 0 and b"workspace" in lib.kgcn_last_error()
    assert call(dotx_ld=d + 2) != 0                                   # rows of dotx must stay 16-byte aligned
    x = torch.zeros((8, 5, 12), device=dev())
    o = torch.zeros((8, 40), device=dev())
    assert lib.kgcn_graph_gather_fwd_ld_f32(ptr(x), 8, 5, 12, ptr(o), 40, current_stream()) == 0
    assert lib.kgcn_graph_gather_fwd_ld_f32(ptr(x), 8, 5, 12, ptr(o), 8, current_stream()) != 0 and b"out_ld" in lib.kgcn_last_error()
    assert lib.kgcn_ragged_blocks(None, 4, 64, ptr(torch.zeros(8, dtype=torch.int32, device=dev())), current_stream()) != 0
    assert lib.kgcn_ragged_num_blocks(0) == 0 and lib.kgcn_ragged_num_blocks(65) == (65 + lib.kgcn_ragged_block_rows() - 1) // lib.kgcn_ragged_block_rows() + 2
    torch.cuda.synchronize()


def test_plain_c_consumer_calls_bspmm_on_the_device():
    """tests/abi_consumer.c (gcc, include/kgcn_hip.h, no Python in the process): struct layout + version checks, then one
    kgcn_bspmm_f32 call on a hand-built 2-graph batch (duplicate entry, empty rows) checked against products computed in C,
    and a refused row_pad batch with its kgcn_last_error() text."""
    from test_abi import run_consumer
    rc, facts, text = run_consumer("bspmm")
    assert rc == 0 and "OK" in facts, text
    assert facts["bspmm_mismatches"].startswith("0 of")
