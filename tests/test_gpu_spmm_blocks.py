"""The aggregation kernel of the ragged-compact batches (csrc/spmm.hip, spmm_block_kernel): a row block of whole molecules is
staged in LDS once (kgcn_csr_batch.block_ptr, built by kgcn_ragged_blocks) and every row gathers there.  It replaces the row-chunk
kernel for these batches, so it must give the SAME BITS (same entry order, same fused multiply-adds) -- checked against the same
container with the block structure removed -- and the fp64 numpy product of kgcn/layers.py:105-116 / kgcn/bspmm_call.py:45."""
import numpy as np
import pytest
import torch

from kgcn_amd._lib import lib, ptr, current_stream, check
from test_gpu_parity import dev, t32
from test_oracle_model import tox21_like_batch

pytestmark = pytest.mark.gpu


def _batch(B, N, seed):
    from kgcn_amd import ragged
    rng = np.random.default_rng(seed)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=B, N=N, F=3, T=2)
    rb = ragged.compact(t32(x), adjs, sizes)
    return rb, sizes, rng


def _dense(csr):
    rp = csr.rowptr.cpu().numpy().astype(np.int64)
    cv = csr.cv.cpu().numpy()[:rp[-1]]
    a = np.zeros((csr.rows, csr.cols), np.float64)
    np.add.at(a, (np.repeat(np.arange(csr.rows), np.diff(rp)), cv[:, 0]), cv[:, 1].view(np.float32).astype(np.float64))
    return a


def _without_blocks(csr):
    from kgcn_amd.batched_csr import BatchedCSR
    return BatchedCSR(csr.rowptr, csr.cv, csr.num_graphs, csr.rows, csr.cols, csr.max_nnz)


def _act(z, code):
    return [z, 1 / (1 + np.exp(-z)), np.maximum(z, 0), np.tanh(z)][code]


def _dact(a, code):
    return [np.ones_like(a), a * (1 - a), (a > 0).astype(np.float64), 1 - a * a][code]


def test_block_table_covers_every_row_with_whole_molecules():
    for B, N, seed in ((300, 50, 1), (7, 12, 2), (40, 132, 3), (1, 9, 4)):
        rb, sizes, _ = _batch(B, N, seed)
        a = rb.adjacency.channels[0]
        bp = a.block_ptr.cpu().numpy()
        gp = rb.graph_ptr.cpu().numpy()
        assert a.transpose().block_ptr is a.block_ptr and a.block_rows_max == lib.kgcn_ragged_block_rows() + N - 1
        assert bp[0] == 0 and bp[-1] == rb.capacity and np.all(np.diff(bp) >= 0) and np.diff(bp).max() <= a.block_rows_max
        assert len(bp) == lib.kgcn_ragged_num_blocks(rb.capacity) + 1
        R = int(gp[-1])
        inside = bp[bp < R]
        assert np.all(np.isin(inside, gp))                              # molecule boundaries only
        for k, b in enumerate(bp[:-1]):
            if b < R:
                assert b >= k * lib.kgcn_ragged_block_rows() and (b == 0 or gp[np.searchsorted(gp, b) - 1] < k * lib.kgcn_ragged_block_rows())


@pytest.mark.parametrize("B,N,d", [(300, 50, 256), (300, 50, 64), (300, 50, 50), (300, 50, 84), (120, 50, 32), (64, 50, 6),
                                    (40, 132, 256), (9, 300, 128), (7, 12, 256), (1, 9, 8)])
def test_forward_aggregation_same_bits_as_the_row_kernel_and_equal_to_fp64(B, N, d):
    rb, sizes, rng = _batch(B, N, 100 + d + N)
    a = rb.adjacency.channels[0]
    plain = _without_blocks(a)
    x = t32(rng.standard_normal((rb.capacity, d)))
    ref = _dense(a) @ x.double().cpu().numpy()
    for act in (0, 1, 2):
        outs = []
        for csr in (a, plain):
            out = torch.full((rb.capacity, d), 7.0, device=dev())
            check(lib.kgcn_bconv_act_f32(csr.desc(), 1, ptr(x), d, rb.capacity * d, 0, d, ptr(out), d, rb.capacity * d, act,
                                         current_stream()))
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), (act, float((outs[0] - outs[1]).abs().max()))
        want = _act(ref, act)
        err = np.abs(outs[0].double().cpu().numpy() - want).max()
        assert err <= 2e-6 * max(1.0, np.abs(want).max()), (act, err)
    # beta = 1: accumulate into the output
    out0 = t32(rng.standard_normal((rb.capacity, d)))
    outs = []
    for csr in (a, plain):
        out = out0.clone()
        check(lib.kgcn_bspmm_f32(csr.desc(), ptr(x), d, rb.capacity * d, d, ptr(out), d, rb.capacity * d, 1.0, current_stream()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert np.abs(outs[0].double().cpu().numpy() - (out0.double().cpu().numpy() + ref)).max() <= 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("B,N,d", [(300, 50, 256), (300, 50, 50), (40, 132, 256), (120, 50, 32), (9, 300, 64)])
@pytest.mark.parametrize("act", [1, 2, 3])
def test_adjoint_with_activation_derivative(B, N, d, act):
    """out = A^T (grad (.) act'(saved output)) -- the backward of an activated GraphConv, kgcn/bspmm_call.py:45"""
    rb, sizes, rng = _batch(B, N, 7 + d + N)
    at = rb.adjacency.channels[0].transpose()
    plain = _without_blocks(at)
    g = t32(rng.standard_normal((rb.capacity, d)))
    aout = t32(_act(rng.standard_normal((rb.capacity, d)), act))
    outs = []
    for csr in (at, plain):
        out = torch.full((rb.capacity, d), 7.0, device=dev())
        check(lib.kgcn_bspmm_dact_f32(csr.desc(), ptr(g), ptr(aout), d, rb.capacity * d, d, act, ptr(out), d, rb.capacity * d,
                                      0.0, current_stream()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    want = _dense(at) @ (g.double().cpu().numpy() * _dact(aout.double().cpu().numpy(), act))
    assert np.abs(outs[0].double().cpu().numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("d", [256, 50])
def test_gin_aggregate_self_term(d):
    """kgcn/layers.py:461-472: eps * x + A x on a ragged-compact batch"""
    rb, sizes, rng = _batch(200, 50, 5 + d)
    a = rb.adjacency.channels[0]
    plain = _without_blocks(a)
    x = t32(rng.standard_normal((rb.capacity, d)))
    eps = t32(np.array([0.37]))
    outs = []
    for csr in (a, plain):
        out = torch.full((rb.capacity, d), 7.0, device=dev())
        check(lib.kgcn_gin_aggregate_f32(csr.desc(), 1, ptr(x), d, ptr(eps), ptr(out), current_stream()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    want = (_dense(a) + 0.37 * np.eye(rb.capacity)) @ x.double().cpu().numpy()
    assert np.abs(outs[0].double().cpu().numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


def test_entries_that_leave_their_block_and_oversized_blocks_are_gathered_from_memory():
    """A hand-built container whose block table cuts THROUGH molecules (entries leave their block), one block longer than the LDS
    capacity of 256 rows and more entries than the LDS entry buffer holds: the result must not depend on the block table."""
    from kgcn_amd.batched_csr import BatchedCSR
    rng = np.random.default_rng(12)
    n, d = 1500, 64
    rows = np.repeat(np.arange(n), 9)
    cols = np.clip(rows + rng.integers(-40, 41, size=rows.size), 0, n - 1)
    vals = rng.standard_normal(rows.size).astype(np.float32)
    rowptr = np.arange(0, 9 * n + 1, 9, dtype=np.int32)
    cv = np.stack([cols.astype(np.int32), vals.view(np.int32)], 1)
    i32 = dict(dtype=torch.int32, device=dev())
    mk = lambda: BatchedCSR(torch.tensor(rowptr, **i32), torch.tensor(cv, **i32), 1, n, n, int(cv.shape[0]))
    plain, blocked = mk(), mk()
    blocked.block_ptr = torch.tensor([0, 10, 10, 75, 400, 401, 900, 1500], **i32)       # 325- and 600-row blocks among them
    blocked.block_rows_max = 600
    x = t32(rng.standard_normal((n, d)))
    outs = []
    for csr in (blocked, plain):
        out = torch.full((n, d), 7.0, device=dev())
        check(lib.kgcn_bspmm_f32(csr.desc(), ptr(x), d, n * d, d, ptr(out), d, n * d, 0.0, current_stream()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    want = _dense(plain) @ x.double().cpu().numpy()
    assert np.abs(outs[0].double().cpu().numpy() - want).max() <= 3e-6 * max(1.0, np.abs(want).max())
