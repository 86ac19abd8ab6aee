"""Ragged-compact batches (kgcn_amd.ragged, csrc/ragged.hip): the layer stack on the valid node rows only must give
what the reference's PADDED formulation gives (kgcn/layers.py GraphConv / GraphDense on all max_node_num rows, BN on
the valid rows :196-210, GraphGather over the padded rows :163-164).  The checker is the numpy oracle on padded tensors."""
import numpy as np
import pytest
import torch

from oracle import kgcn_oracle as K
from test_gpu_parity import close, dev, t32
from test_oracle_model import tox21_like_batch

pytestmark = pytest.mark.gpu


def _block_diagonal_reference(adjs, sizes, capacity):
    """numpy: (rowptr, col, val) of the block-diagonal CSR of the valid blocks, entries in stored order."""
    gp = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(sizes, out=gp[1:])
    rows, cols, vals = [], [], []
    for b, chans in enumerate(adjs):
        idx, val, _ = chans[0]
        idx = np.asarray(idx).reshape(-1, 2)
        rows.append(idx[:, 0] + gp[b]); cols.append(idx[:, 1] + gp[b]); vals.append(np.asarray(val, np.float32))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.argsort(r, kind="stable")
    r, c, v = r[order], c[order], v[order]
    rowptr = np.zeros(capacity + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=capacity), out=rowptr[1:])
    return gp, rowptr, c, v


@pytest.mark.parametrize("B,N,F", [(7, 12, 5), (300, 50, 81), (1, 9, 4)])
def test_compact_layout_bit_exact(B, N, F):
    from kgcn_amd import ragged
    rng = np.random.default_rng(B)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=B, N=N, F=F, T=2)
    rb = ragged.compact(t32(x), adjs, sizes)
    R = int(sizes.sum())
    assert rb.rows == R and rb.capacity >= R + 1 and rb.capacity % 4 == 0
    gp, rowptr, col, val = _block_diagonal_reference(adjs, sizes, rb.capacity)
    assert np.array_equal(rb.graph_ptr.cpu().numpy(), gp)
    assert int(rb.row_count.item()) == R
    a = rb.adjacency.channels[0]
    assert (a.num_graphs, a.rows, a.cols) == (1, rb.capacity, rb.capacity)
    assert np.array_equal(a.rowptr.cpu().numpy(), rowptr)
    E = int(rowptr[-1])
    cv = a.cv.cpu().numpy()[:E]
    assert np.array_equal(cv[:, 0], col) and np.array_equal(cv[:, 1].view(np.float32), val)
    # A^T container: same entries, transposed
    at = a.transpose()
    rpt = at.rowptr.cpu().numpy()
    cvt = at.cv.cpu().numpy()[:E]
    dense = np.zeros((rb.capacity, rb.capacity), np.float64)
    rr = np.repeat(np.arange(rb.capacity), np.diff(rowptr))
    np.add.at(dense, (rr, col), val)
    dense_t = np.zeros_like(dense)
    rt = np.repeat(np.arange(rb.capacity), np.diff(rpt))
    np.add.at(dense_t, (rt, cvt[:, 0]), cvt[:, 1].view(np.float32))
    assert np.array_equal(dense.T, dense_t)
    # features: valid rows stacked, zeros behind
    f = rb.features[0].cpu().numpy()
    want = np.zeros((rb.capacity, F), np.float32)
    for b in range(B):
        want[gp[b]:gp[b + 1]] = x[b, :sizes[b]]
    assert np.array_equal(f, want)
    # round trip to the padded layout
    back = rb.expand(rb.features, fill="zero").cpu().numpy()
    assert np.array_equal(back, x.astype(np.float32))


def test_compact_rejects_entries_beyond_the_true_size():
    from kgcn_amd import ragged
    rng = np.random.default_rng(3)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=5, N=10, F=3, T=2)
    sizes = sizes.copy()
    sizes[0] -= 1                                  # graph 0 now has entries on its last (no longer valid) node
    with pytest.raises(ValueError, match="beyond enabled_node_nums"):
        ragged.compact(t32(x), adjs, sizes)
    with pytest.raises(ValueError, match="capacity"):
        ragged.compact(t32(x), adjs, sizes + (np.arange(5) == 0), capacity=8)


@pytest.mark.parametrize("d", [50, 256, 7])
def test_ragged_gather_matches_padded_reduce_sum(d):
    """GraphGather on the compact layout = reduce_sum over the PADDED rows when every padded row holds the padding
    representative's value (quirk Q4); backward: valid rows get dout[b], the representative row sum_b (N - n_b) dout[b]."""
    from kgcn_amd import layers, ragged
    rng = np.random.default_rng(d)
    B, N = 37, 20
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=B, N=N, F=3, T=2)
    rb = ragged.compact(None, adjs, sizes)
    h = rng.standard_normal((rb.capacity, d)).astype(np.float32)
    th = t32(h).reshape(1, rb.capacity, d).requires_grad_(True)
    out = layers.GraphGather()(th, ragged=rb)
    gp = rb.graph_ptr.cpu().numpy()
    padded = np.repeat(h[rb.pad_row][None, None, :], B, 0).repeat(N, 1).astype(np.float64)
    for b in range(B):
        padded[b, :sizes[b]] = h[gp[b]:gp[b + 1]]
    close(out, K.gather_fwd(padded), atol=2e-5, what="ragged gather")
    g = rng.standard_normal((B, d)).astype(np.float32)
    out.backward(t32(g))
    want = np.zeros((rb.capacity, d))
    for b in range(B):
        want[gp[b]:gp[b + 1]] = g[b]
    want[rb.pad_row] = ((N - sizes)[:, None] * g.astype(np.float64)).sum(0)
    close(th.grad[0], want, atol=1e-5, rel=1e-6, what="ragged gather backward")
    close(rb.expand(th.detach()), padded, atol=0, what="expand with the padding representative")


@pytest.mark.parametrize("channels", [1, 2])
def test_model_py_network_ragged_equals_padded_oracle(channels):
    """example_model/model.py's network (3 x GraphConv, BN on the valid rows, GraphDense, gather, Dense) on a padded batch
    with true sizes: ragged-compact product path vs the padded product path vs (through test_gpu_model) the oracle --
    logits and every parameter gradient."""
    from kgcn_amd import models
    rng = np.random.default_rng(5 + channels)
    B, N, F = 40, 16, 6
    x, adjs, labels, mask, _, sizes = tox21_like_batch(rng, B=B, N=N, F=F, T=2)
    if channels == 2:                                   # a second channel: the transposed pattern with other values
        adjs = [[a[0], (np.asarray(a[0][0])[:, ::-1].copy(), (np.asarray(a[0][1]) * 0.5).astype(np.float32), a[0][2])]
                for a in adjs]
    lab = np.eye(2)[rng.integers(0, 2, B)]
    res = {}
    for ragged_mode in (False, True):
        torch.manual_seed(0)
        model = models.GCN(channels, 2, ragged=ragged_mode).to(dev())
        tx = t32(x).requires_grad_(True)
        en = torch.as_tensor(sizes)
        model(tx, adjs, enabled_node_nums=en)
        with torch.no_grad():                            # non-trivial biases / BN parameters
            gen = torch.Generator(device="cpu").manual_seed(1)
            for p_ in model.parameters():
                if p_.dim() == 1 or p_.shape[0] == 1:
                    p_.copy_(torch.randn(p_.shape, generator=gen).to(p_.device) * 0.1 + (1.0 if p_ is model.bn.gamma else 0.0))
        logits = model(tx, adjs, enabled_node_nums=en)
        cost, _ = models.masked_softmax_ce(logits, t32(lab), t32(mask))
        cost.backward()
        res[ragged_mode] = (logits.detach().cpu().numpy(), tx.grad.cpu().numpy(),
                            [p_.grad.cpu().numpy() for p_ in model.parameters()])
    close(res[True][0], res[False][0], atol=2e-5, what="logits ragged vs padded")
    close(res[True][1], res[False][1], atol=1e-6, rel=2e-5, what="d features ragged vs padded")
    for i, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        close(a, b, atol=2e-6, rel=2e-5, what="parameter %d gradient ragged vs padded" % i)


def test_static_ragged_batch_equals_compact_and_replays():
    """Device-side assembly of a ragged-compact mini-batch from a resident dataset (fixed capacity, fixed addresses) equals
    compact() of the same padded batch; the hipGraph-captured train step on it follows the eager padded step."""
    from kgcn_amd import data_util as D, models, ragged, train
    rng = np.random.default_rng(9)
    G, N, F, T, B = 300, 20, 7, 3, 32
    x, adjs, labels, mask, mask_label, sizes = tox21_like_batch(rng, B=G, N=N, F=F, T=T)
    sizes = np.maximum(sizes, 1); x[-1, :1] = 0
    flat = D.FlatAdjacency.from_coo_list([a[0] for a in adjs], n_nodes=N)
    ds = D.DeviceGraphDataset([flat], x.astype(np.float32), device=dev(), sizes=np.where(np.arange(G) == G - 1, 0, sizes))
    sizes = ds.sizes
    srb = ds.static_ragged_batch(B)
    assert srb.capacity % 64 == 0 and srb.capacity <= B * N + 64
    for trial in range(3):
        idx = rng.permutation(G)[:B - (trial == 2) * 5]            # the last one is a short batch (dummy graphs)
        srb.load(idx)
        assert int(srb.status.item()) == 0
        pad_adj, pad_x = ds.batch(idx, B)
        en = np.zeros(B, np.int64); en[:len(idx)] = sizes[idx]
        ref = ragged.compact(pad_x, pad_adj, en, capacity=srb.capacity)
        R = int(en.sum())
        assert np.array_equal(srb.ragged.graph_ptr.cpu().numpy(), ref.graph_ptr.cpu().numpy())
        for a, b in ((srb.ragged.adjacency.channels[0], ref.adjacency.channels[0]),
                     (srb.ragged.adjacency.channels[0].transpose(), ref.adjacency.channels[0].transpose())):
            assert np.array_equal(a.rowptr.cpu().numpy(), b.rowptr.cpu().numpy())
            E = int(a.rowptr[-1])
            assert np.array_equal(a.cv.cpu().numpy()[:E], b.cv.cpu().numpy()[:E])
        assert np.array_equal(srb.features.cpu().numpy(), ref.features.detach().cpu().numpy())
        assert srb.ragged.rows == R
    # captured step on the static ragged batch vs eager padded steps, same batches, same initial weights
    lab_d, ml_d = t32(labels), t32(mask_label)
    batches = [rng.permutation(G)[:B] for _ in range(4)]

    def fresh():
        torch.manual_seed(3)
        m = models.MultitaskGCN(1, T, ragged=True).to(dev())
        a0, x0 = ds.batch(batches[0], B)
        m(x0, a0, enabled_node_nums=torch.as_tensor(sizes[batches[0]]))
        return m
    m_e, m_g = fresh(), fresh()
    ones = torch.ones(B, device=dev())
    opt_e = train.TFAdam(m_e.parameters(), lr=1e-2)
    for b in batches:
        a, xb = ds.batch(b, B)
        it = torch.as_tensor(b, device=dev())
        train.train_step(m_e, opt_e, lambda lg, lb, mk: models.masked_sigmoid_ce(lg, lb, mk, ml_d[it]), xb, a, lab_d[it], ones,
                         enabled_node_nums=torch.as_tensor(sizes[b]))
    opt_g = train.TFAdam(m_g.parameters(), lr=1e-2, capturable=True)
    lab_s, ml_s = torch.zeros((B, T), device=dev()), torch.zeros((B, T), device=dev())
    srb.load(batches[0])
    step = train.GraphedTrainStep(m_g, opt_g, lambda lg, lb, mk: models.masked_sigmoid_ce(lg, lb, mk, ml_s), srb, lab_s, ones)
    for b in batches:
        it = torch.as_tensor(b, device=dev())
        srb.load(b); lab_s.copy_(lab_d[it]); ml_s.copy_(ml_d[it])
        step.replay()
    torch.cuda.synchronize()
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        close(pg, pe.detach().cpu().numpy(), atol=2e-5, rel=1e-4, what="parameters after 4 steps: captured ragged vs eager")


def test_static_ragged_batch_with_augmented_feature_rows():
    """StaticRaggedBatch(augmented_features=True): the rows are assembled as [x | 1 | 0] (kgcn_ragged_compact_rows_aug_f32) -- bit for
    bit what kgcn_ragged_compact_rows_f32 + kgcn_augment_ones_f32 write --, `.features` is a view of them, an aggregate-first GraphConv
    reads the buffer where it lies (no augment_ones pass) and the captured train step lands on the SAME parameters, bit for bit, as the
    one on the plain batch."""
    from kgcn_amd import data_util as D, models, ops, train
    rng = np.random.default_rng(19)
    G, N, F, T, B = 400, 20, 7, 3, 160
    x, adjs, labels, mask, mask_label, sizes = tox21_like_batch(rng, B=G, N=N, F=F, T=T)
    sizes = np.maximum(sizes, 1)
    flat = D.FlatAdjacency.from_coo_list([a[0] for a in adjs], n_nodes=N)
    ds = D.DeviceGraphDataset([flat], x.astype(np.float32), device=dev(), sizes=sizes)
    plain, aug = ds.static_ragged_batch(B), ds.static_ragged_batch(B, augmented_features=True)
    assert aug.capacity == plain.capacity and tuple(aug.features.shape) == tuple(plain.features.shape)
    assert aug.features._kgcn_aug.shape == (aug.capacity, 8) and aug.features._kgcn_aug.data_ptr() == aug.features.data_ptr()
    for trial in range(3):
        idx = rng.permutation(G)[:B - (trial == 2) * 9]            # the last one is a short batch (dummy graphs)
        plain.load(idx); aug.load(idx)
        assert np.array_equal(aug.features.cpu().numpy(), plain.features.cpu().numpy())
        want = ops.augment_ones(plain.features[0], 8).cpu().numpy()
        assert np.array_equal(aug.features._kgcn_aug.cpu().numpy(), want)
    assert not models.wants_augmented_features(models.GIN(2), F)
    lab_d, ml_d = t32(labels), t32(mask_label)
    batches = [rng.permutation(G)[:B] for _ in range(3)]
    ones = torch.ones(B, device=dev())
    out = []
    seen = {"augment": 0}
    real = ops.augment_ones
    for sb in (plain, aug):
        torch.manual_seed(3)
        m = models.MultitaskGCN(1, T, ragged=True).to(dev())
        m.conv1 = type(m.conv1)(64, 1, activation="sigmoid")     # 7 + 1 -> 8 < 64: aggregate-first (>= 1,024 rows)
        assert sb.capacity >= 1024
        assert models.wants_augmented_features(m, F)
        sb.load(batches[0])
        en_s = sb.add_table(torch.as_tensor(sizes.astype(np.int32), device=dev()))
        sb.load(batches[0])
        m(sb.features, sb.adjacency, enabled_node_nums=en_s)
        opt = train.TFAdam(m.parameters(), lr=1e-2, capturable=True)
        lab_s, ml_s = torch.zeros((B, T), device=dev()), torch.zeros((B, T), device=dev())

        def counting(x2d, width):
            seen["augment"] += int(width == 8)       # (conv2, 64 -> 256, is aggregate-first too: its operand is made by augment_ones)
            return real(x2d, width)
        ops.augment_ones = counting
        try:
            step = train.GraphedTrainStep(m, opt, lambda lg, lb, mk: models.masked_sigmoid_ce(lg, lb, mk, ml_s), sb, lab_s, ones,
                                          enabled_node_nums=en_s)
        finally:
            ops.augment_ones = real
        if sb is aug:
            assert seen["augment"] == 0, "the augmented rows were not read where they lie"
        else:
            assert seen["augment"] > 0
            seen["augment"] = 0
        for b in batches:
            it = torch.as_tensor(b, device=dev())
            sb.load(b); lab_s.copy_(lab_d[it]); ml_s.copy_(ml_d[it])
            step.replay()
        torch.cuda.synchronize()
        out.append([p_.detach().cpu().numpy().copy() for p_ in m.parameters()])
        del step
    for a, b in zip(*out):
        assert np.array_equal(a, b), "parameters after 3 captured steps differ between the plain and the augmented batch"


@pytest.mark.parametrize("B,N,din,dout,C,act", [(40, 50, 81, 256, 1, "sigmoid"), (64, 20, 6, 40, 2, None), (30, 40, 30, 50, 1, "relu"),
                                                (64, 20, 7, 64, 6, "tanh")])
def test_graphconv_aggregate_first_equals_contract_first(B, N, din, dout, C, act):
    """A (X W + 1 b) evaluated as (A [X | 1]) [W ; b] (layers.aggregate_first, taken when din + 1 < dout): forward, d inputs,
    dW and dbias against the oracle's contract-first formulation (kgcn/layers.py:105-116) with NON-ZERO biases, graphs with
    empty adjacency rows (padded nodes: rowsum(A) = 0, so the bias must NOT reach them) and several channels; and against
    the product's own contract-first route."""
    from kgcn_amd import layers
    rng = np.random.default_rng(B + din)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=B, N=N, F=din, T=2)
    if C > 1:
        adjs = [[(a[0][0], (np.asarray(a[0][1]) * (0.5 + c)).astype(np.float32), a[0][2]) if c % 2 == 0 else
                 (np.asarray(a[0][0])[:, ::-1].copy(), (np.asarray(a[0][1]) * (0.5 + c)).astype(np.float32), a[0][2])
                 for c in range(C)] for a in adjs]
    g = rng.standard_normal((B, N, dout)).astype(np.float32)
    res = {}
    for first in (True, False):
        layers.aggregate_first = first
        try:
            torch.manual_seed(0)
            layer = layers.GraphConv(dout, C, activation=act)
            tx = t32(x).requires_grad_(True)
            layer.build(tx.shape, dev())
            with torch.no_grad():
                for c in range(C):
                    layer.bias[c].copy_(t32(np.random.default_rng(c).standard_normal((1, dout)) * 0.3))
            out = layer(tx, adj=adjs)
            out.backward(t32(g))
            res[first] = (out.detach().cpu().numpy(), tx.grad.cpu().numpy(), [p.grad.cpu().numpy() for p in layer.w],
                          [p.grad.cpu().numpy() for p in layer.bias], [p.detach().cpu().numpy() for p in layer.w],
                          [p.detach().cpu().numpy() for p in layer.bias])
        finally:
            layers.aggregate_first = True
    out, dx, dw, db, w, b = res[True]
    pre = K.graphconv_fwd(x, adjs, w, b)
    if act == "sigmoid":
        ref, dpre = 1 / (1 + np.exp(-pre)), None
        dpre = g * ref * (1 - ref)
    elif act == "relu":
        ref = np.maximum(pre, 0); dpre = g * (out > 0)
    elif act == "tanh":
        ref = np.tanh(pre); dpre = g * (1 - ref * ref)
    else:
        ref, dpre = pre, g
    close(out, ref, atol=2e-6, rel=1e-6, what="aggregate-first fwd")
    assert np.all(out[-1] == (0.5 if act == "sigmoid" else 0.0)), "rows of an empty graph must not see the bias"
    rdx, rdw, rdb = K.graphconv_bwd(x, adjs, w, b, dpre)
    close(dx, rdx, atol=1e-6, rel=3e-6, what="aggregate-first d inputs")
    for c in range(C):
        close(dw[c], rdw[c], atol=1e-6, rel=3e-6, what="aggregate-first dW channel %d" % c)
        close(db[c], rdb[c], atol=1e-6, rel=3e-6, what="aggregate-first dbias channel %d" % c)
    close(out, res[False][0], atol=3e-6, rel=1e-6, what="aggregate-first vs contract-first fwd")
    close(dx, res[False][1], atol=2e-6, rel=5e-6, what="aggregate-first vs contract-first d inputs")


def test_tiny_ragged_batch_takes_the_unfused_route():
    """A ragged-compact batch of fewer than 32 rows is a one-graph batch whose shape would fit the fused GraphConv kernels, but
    its container has no row-padded copy: it must go through the dense + row-chunk aggregation route (found by tools/fuzz_gpu.py)."""
    from kgcn_amd import layers, ragged
    rng = np.random.default_rng(0)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=3, N=8, F=5, T=2)
    rb = ragged.compact(t32(x), adjs, sizes)
    assert rb.capacity <= 32
    conv = layers.GraphConv(7, 1, activation="sigmoid")
    out = conv(rb.features, adj=rb)
    ref = 1 / (1 + np.exp(-K.graphconv_fwd_fast(x, adjs, [conv.w[0].detach().cpu().numpy()], [conv.bias[0].detach().cpu().numpy()])))
    valid = (np.arange(8)[None, :] < sizes[:, None])[:, :, None]
    close(rb.expand(out, fill="zero").cpu().numpy() * valid, ref * valid, atol=2e-6, what="tiny ragged GraphConv")


@pytest.mark.parametrize("T", [1, 63, 1024, 1025, 4096, 8192, 8193, 20000])
def test_plan_scans_of_small_and_multi_block_selections(T):
    """kgcn_ragged_plan: graph_ptr / entry_ptr = exclusive scans of the valid rows / stored entries of the selected graphs (with
    dummy selections and sizes beyond the padded size), from one scan block to many."""
    from kgcn_amd import data_util as D
    from kgcn_amd._lib import lib, ptr, check, current_stream
    rng = np.random.default_rng(T)
    G, N = 500, 12
    chans, _ = D.build_adjs({"dense_adj": (rng.random((G, N, N)) < 0.2).astype(np.int64), "max_node_num": N})
    src = D.DeviceGraphDataset(chans, None, device=dev()).channels[0]
    sizes = rng.integers(-2, N + 4, size=G).astype(np.int32)
    sel = rng.integers(-1, G, size=T).astype(np.int32)
    rp = src.rowptr.cpu().numpy().astype(np.int64)
    n = np.where(sel >= 0, np.clip(sizes[np.maximum(sel, 0)], 0, N), 0)
    e = np.where(sel >= 0, rp[np.maximum(sel, 0) * N + n] - rp[np.maximum(sel, 0) * N], 0)
    i32 = dict(dtype=torch.int32, device=dev())
    gp, ep = torch.full((T + 1,), -7, **i32), torch.full((T + 1,), -7, **i32)
    ws = torch.empty(max(lib.kgcn_ragged_workspace_bytes(T), 8) // 4, **i32)
    check(lib.kgcn_ragged_plan(src.desc(), ptr(torch.tensor(sizes, **i32)), ptr(torch.tensor(sel, **i32)), T, ptr(gp), ptr(ep),
                               ptr(ws), ws.numel() * 4, current_stream()))
    assert np.array_equal(gp.cpu().numpy(), np.concatenate([[0], np.cumsum(n)]))
    assert np.array_equal(ep.cpu().numpy(), np.concatenate([[0], np.cumsum(e)]))


def test_compact_checks_an_explicit_capacity_before_writing_and_refuses_differentiable_values():
    """(1) sizes on the device + an explicit capacity: the row count is read back BEFORE the device kernels write rows
    [0, R] (they are not bounded by the capacity); (2) adjacency values as differentiable inputs have no d values path on the compact
    layout: compact() raises, a ragged model falls back to the padded layout (the gradient still arrives)."""
    from kgcn_amd import ragged, models
    from kgcn_amd.batched_csr import BatchedAdjacency
    rng = np.random.default_rng(8)
    x, adjs, _, _, _, sizes = tox21_like_batch(rng, B=6, N=10, F=3, T=2)
    R = int(sizes.sum())
    with pytest.raises(ValueError, match="capacity"):
        ragged.compact(t32(x), adjs, torch.as_tensor(sizes, device=dev()), capacity=R, check=False)
    a = BatchedAdjacency.from_adjs(adjs, device=dev())
    vals = [ch.values.clone().requires_grad_(True) for ch in a.channels]
    av = a.with_values(vals)
    with pytest.raises(ValueError, match="differentiable adjacency values"):
        ragged.compact(t32(x), av, sizes)
    model = models.GCN(1, ragged=True).to(dev())
    out = model(t32(x), av, enabled_node_nums=torch.as_tensor(sizes))
    out.sum().backward()
    assert vals[0].grad is not None and float(vals[0].grad.abs().max()) > 0
