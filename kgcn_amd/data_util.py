"""Host-side input producers of the hot path, vectorised over the WHOLE dataset.

The reference builds adjacency graph by graph in Python loops (kgcn/data_util.py:40-45 dense ->
COO, :58-73 powers, :76-122 degree split, :125-140 Kipf normalisation, :396-420 build_data) and
assembles every mini-batch entry by entry (kgcn/feed.py:112-133).  Here one adjacency channel of
the whole dataset is four flat arrays (graph, row, col, val) in the reference's stored-entry
order; every transformation is a handful of numpy array operations over all graphs at once, and a
mini-batch (including the reference's empty dummy graphs that pad a short batch,
kgcn/feed.py:123-126) is a slice-gather of those arrays straight into a device BatchedCSR.

Results are pinned bit-exactly against the reference's own outputs for the shipped datasets
(tests/test_product_loaders.py, fixtures tests/golden/g2*, g3*, g4*).
"""
import numpy as np
import scipy.sparse as sp

from . import _lib
from .batched_csr import BatchedAdjacency, BatchedCSR


class FlatAdjacency:
    """One adjacency channel of G graphs: entries sorted by graph, stored order inside a graph."""

    def __init__(self, graph, row, col, val, num_graphs, n_rows, n_cols=None):
        self.graph = np.asarray(graph, np.int64)
        self.row = np.asarray(row, np.int32)
        self.col = np.asarray(col, np.int32)
        self.val = np.asarray(val, np.float32)
        self.num_graphs = int(num_graphs)
        self.n_rows = int(n_rows)
        self.n_cols = int(n_rows if n_cols is None else n_cols)
        self.ptr = np.zeros(self.num_graphs + 1, np.int64)
        np.cumsum(np.bincount(self.graph, minlength=self.num_graphs), out=self.ptr[1:])

    # -- construction -------------------------------------------------------------------------
    @classmethod
    def from_dense(cls, dense_adj):
        """dense [G,N,N] -> COO in row-major order of the non-zeros (kgcn/data_util.py:40-45)."""
        a = np.asarray(dense_adj)
        g, r, c = np.nonzero(a)
        return cls(g, r, c, a[g, r, c].astype(np.float32), a.shape[0], a.shape[1], a.shape[2])

    @classmethod
    def from_coo_list(cls, mats, n_nodes=None):
        """mats[g] = (idx [nnz,2], val [nnz], shape) -- the "adj" key of a .jbl file."""
        gs, rs, cs, vs = [], [], [], []
        n = 0
        for g, m in enumerate(mats):
            idx = np.asarray(m[0]).reshape(-1, 2)
            gs.append(np.full(idx.shape[0], g, np.int64))
            rs.append(idx[:, 0])
            cs.append(idx[:, 1])
            vs.append(np.asarray(m[1], np.float32).reshape(-1))
            n = max(n, int(m[2][0]))
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        return cls(cat(gs, np.int64), cat(rs, np.int32), cat(cs, np.int32), cat(vs, np.float32),
                   len(mats), n_nodes if n_nodes is not None else n)

    # -- views ----------------------------------------------------------------------------------
    def to_list(self, shape=None):
        """Back to the reference's per-graph layout adjs[g] = (idx int32 [nnz,2], val f32, shape)."""
        shp = [self.n_rows, self.n_cols] if shape is None else shape
        idx = np.stack([self.row, self.col], axis=1).astype(np.int32)
        return [(idx[self.ptr[g]:self.ptr[g + 1]], self.val[self.ptr[g]:self.ptr[g + 1]], shp)
                for g in range(self.num_graphs)]

    def key(self):
        return (self.graph * self.n_rows + self.row) * self.n_cols + self.col

    def batch(self, batch_idx, batch_size=None, device="cuda"):
        """Mini-batch as a device BatchedCSR: graphs `batch_idx` in order, padded with empty graphs
        up to batch_size (kgcn/feed.py:112-126)."""
        batch_idx = np.asarray(batch_idx, np.int64)
        nb = batch_idx.shape[0]
        T = nb if batch_size is None else int(batch_size)
        lo, hi = self.ptr[batch_idx], self.ptr[batch_idx + 1]
        cnt = hi - lo
        total = int(cnt.sum())
        # concatenated ranges lo[i]..hi[i]
        start = np.zeros(nb + 1, np.int64)
        np.cumsum(cnt, out=start[1:])
        sel = np.arange(total, dtype=np.int64) - np.repeat(start[:-1], cnt) + np.repeat(lo, cnt)
        newg = np.repeat(np.arange(nb, dtype=np.int64), cnt)
        return BatchedCSR.from_arrays(newg, self.row[sel], self.col[sel], self.val[sel], T, self.n_rows,
                                      self.n_cols, device=device)


# -------------------------------------------------------------------------------------------------
# transformations (whole dataset at once)
# -------------------------------------------------------------------------------------------------
def normalize_adj(fa):
    """Kipf normalisation (kgcn/data_util.py:125-140): binarise positive values, scale by
    D^-1/2 on both sides with D = COLUMN sums (0 -> 1); float32 arithmetic as the reference's
    expression evaluates (two multiplications by the float32 reciprocal square root).  Pattern in
    canonical row-major order with duplicates summed; explicit zeros are kept."""
    v = fa.val.copy()
    v[v > 0] = 1
    key = fa.key()
    order = np.argsort(key, kind="stable")
    ks, vs = key[order], v[order]
    first = np.ones(ks.shape[0], bool)
    first[1:] = ks[1:] != ks[:-1]
    seg = np.cumsum(first) - 1
    vsum = np.zeros(int(first.sum()), np.float32)
    np.add.at(vsum, seg, vs)
    ku = ks[first]
    col = (ku % fa.n_cols).astype(np.int32)
    gr = ku // fa.n_cols
    row = (gr % fa.n_rows).astype(np.int32)
    g = gr // fa.n_rows
    deg = np.zeros(fa.num_graphs * fa.n_cols, np.float32)
    np.add.at(deg, g * fa.n_cols + col, vsum)
    deg[deg == 0] = 1
    recip = (1.0 / np.sqrt(deg)).astype(np.float32)
    # A / sqrt(d)[:,None] / sqrt(d): the row factor is indexed by the ROW index into the same
    # (column-sum) degree vector, exactly like the reference's broadcasting does
    out = (vsum * recip[g * fa.n_cols + row]) * recip[g * fa.n_cols + col]
    return FlatAdjacency(g, row, col, out.astype(np.float32), fa.num_graphs, fa.n_rows, fa.n_cols)


def split_adj(fa, min_deg=1, max_deg=5):
    """Degree split (kgcn/data_util.py:76-122): off-diagonal entries go to the channel of their
    row's entry count (clamped at max_deg), diagonal entries to the last channel.  Every channel
    of every graph starts with an explicit (0,0)->0.0 entry, dropped only when the channel's
    first real entry is itself at (0,0)."""
    nch = (max_deg - min_deg + 1) + 1
    G, N = fa.num_graphs, fa.n_rows
    deg = np.bincount(fa.graph * N + fa.row, minlength=G * N)
    ch = np.minimum(deg[fa.graph * N + fa.row], max_deg) - min_deg
    ch = np.where(fa.row == fa.col, nch - 1, ch)
    out = []
    pos = np.arange(fa.graph.shape[0])
    for k in range(nch):
        m = ch == k
        g, r, c, v, p = fa.graph[m], fa.row[m], fa.col[m], fa.val[m], pos[m]
        # first real entry of every graph in this channel
        has = np.zeros(G, bool)
        has[g] = True
        firstpos = np.full(G, np.iinfo(np.int64).max)
        np.minimum.at(firstpos, g, p)
        first_is_origin = np.zeros(G, bool)
        sel = has.nonzero()[0]
        fp = firstpos[sel]
        first_is_origin[sel] = (fa.row[fp] == 0) & (fa.col[fp] == 0)
        dummy_g = np.nonzero(~first_is_origin)[0]            # graphs that keep the dummy entry
        gg = np.concatenate([dummy_g, g])
        rr = np.concatenate([np.zeros(dummy_g.shape[0], np.int32), r])
        cc = np.concatenate([np.zeros(dummy_g.shape[0], np.int32), c])
        vv = np.concatenate([np.zeros(dummy_g.shape[0], np.float32), v])
        pp = np.concatenate([np.full(dummy_g.shape[0], -1, np.int64), p])   # dummy first
        order = np.lexsort((pp, gg))
        out.append(FlatAdjacency(gg[order], rr[order], cc[order], vv[order], G, N, fa.n_cols))
    return out


def high_order_adj(fa, order):
    """Pattern of A^order with unit values, indices sorted (kgcn/data_util.py:58-73), computed on
    the block-diagonal matrix of the whole dataset."""
    if order <= 1:
        return fa
    n = fa.num_graphs * fa.n_rows
    a = sp.csr_matrix((fa.val, (fa.graph * fa.n_rows + fa.row, fa.graph * fa.n_rows + fa.col)), shape=(n, n))
    b = a
    for _ in range(order - 1):
        b = b.dot(a)
    b = b.tocoo()
    k = np.sort(b.row.astype(np.int64) * n + b.col)
    r, c = k // n, k % n
    return FlatAdjacency(r // fa.n_rows, r % fa.n_rows, c % fa.n_rows, np.ones(k.shape[0], np.float32),
                         fa.num_graphs, fa.n_rows, fa.n_cols)


def build_adjs(data, normalize_adj_flag=False, split_adj_flag=False, order=1):
    """Dataset dict (.jbl content) -> list of adjacency channels (FlatAdjacency), following
    build_data (kgcn/data_util.py:396-420): "dense_adj" | "adj" (+ "max_node_num") |
    "multi_dense_adj"; powers 1..order are separate channels; then split, then normalise.
    Returns (channels, enabled_node_nums)."""
    if "multi_dense_adj" in data:
        mats = data["multi_dense_adj"]
        nch = len(mats[0])
        chans = [FlatAdjacency.from_dense(np.stack([np.asarray(m[c]) for m in mats])) for c in range(nch)]
        enabled = np.array([max(len(x) for x in m) for m in mats], np.int32)
    else:
        max_n = int(data["max_node_num"])
        if "adj" in data:
            base = FlatAdjacency.from_coo_list(data["adj"], n_nodes=max_n)
            enabled = np.array([int(m[2][0]) for m in data["adj"]], np.int32)
        else:
            dense = np.asarray(data["dense_adj"])
            base = FlatAdjacency.from_dense(dense)
            enabled = np.full(dense.shape[0], dense.shape[1], np.int32)
            base.n_rows = base.n_cols = max_n                # align_size, kgcn/data_util.py:30-37
        chans = [high_order_adj(base, o) for o in range(1, order + 1)]
    if split_adj_flag:
        chans = [c for ch in chans for c in split_adj(ch)]
    if normalize_adj_flag:
        chans = [normalize_adj(ch) for ch in chans]
    return chans, enabled


def batch_adjacency(channels, batch_idx, batch_size=None, device="cuda"):
    """All channels of one mini-batch as a BatchedAdjacency (what GraphConv/GINAggregate take)."""
    return BatchedAdjacency([c.batch(batch_idx, batch_size, device) for c in channels])


def batch_features(features, batch_idx, batch_size=None, device="cuda"):
    """kgcn/feed.py:127-133: float32 features of the batch, zero rows for the padding graphs."""
    import torch
    batch_idx = np.asarray(batch_idx, np.int64)
    T = batch_idx.shape[0] if batch_size is None else int(batch_size)
    out = np.zeros((T,) + tuple(features.shape[1:]), np.float32)
    out[:batch_idx.shape[0]] = features[batch_idx]
    return torch.from_numpy(out).to(device)


class DeviceGraphDataset:
    """The whole dataset resident in HBM: every adjacency channel as ONE batched-CSR
    container over all G graphs (plus, built lazily, its transpose and the row-padded copies the fused
    kernels read), features as one [G, N, F] tensor.  batch() assembles a mini-batch on the device:
    adjacency by kgcn_csr_gather_graphs (segmented copies, no host loops, no per-step upload of the
    adjacency like kgcn/core.py:267-269 does), features by one index_select; short batches are padded
    with empty dummy graphs / zero feature rows exactly like kgcn/feed.py:112-133."""

    def __init__(self, channels, features=None, device="cuda", sizes=None):
        """sizes (optional): true node count of every graph (the reference's enabled_node_nums, kgcn/feed.py:148-151) --
        enables static_ragged_batch(): batches assembled directly in the ragged-compact layout (kgcn_amd.ragged)."""
        import torch
        self.num_graphs = channels[0].num_graphs
        all_idx = np.arange(self.num_graphs)
        self.sizes = self.sizes_dev = None
        if sizes is not None:
            self.sizes = np.asarray(sizes, np.int64).reshape(-1)
            if self.sizes.shape[0] != self.num_graphs:
                raise ValueError("sizes has %d entries for %d graphs" % (self.sizes.shape[0], self.num_graphs))
            n = channels[0].n_rows
            if self.sizes.min(initial=0) < 0 or self.sizes.max(initial=0) > n:
                raise ValueError("sizes must lie in [0, %d]" % n)
            for c in channels:                      # no stored entry may touch a node >= size (checked once, on the host)
                lim = self.sizes[c.graph]
                if c.graph.size and ((c.row >= lim).any() or (c.col >= lim).any()):
                    raise ValueError("an adjacency entry touches a node beyond its graph's size")
            self.sizes_dev = torch.from_numpy(self.sizes.astype(np.int32)).to(device)
        self.channels = [c.batch(all_idx, device=device) for c in channels]
        self.features = None if features is None else \
            torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(device)

    def batch(self, batch_idx, batch_size=None):
        import torch
        batch_idx = np.asarray(batch_idx, np.int64).reshape(-1)
        T = batch_idx.shape[0] if batch_size is None else int(batch_size)
        sel = np.full(T, -1, np.int64)
        sel[:batch_idx.shape[0]] = batch_idx
        adj = BatchedAdjacency([c.gather(sel) for c in self.channels])
        if self.features is None:
            return adj, None
        idx = torch.from_numpy(batch_idx).to(self.features.device)
        feat = self.features.new_zeros((T,) + tuple(self.features.shape[1:]))
        feat[:batch_idx.shape[0]] = self.features.index_select(0, idx)
        return adj, feat

    def static_batch(self, batch_size, fused=True):
        """Fixed-address batch buffers for hipGraph replay (kgcn_amd.train.GraphedTrainStep)."""
        return StaticBatch(self, batch_size, fused)

    def static_ragged_batch(self, batch_size, capacity=None, augmented_features=False):
        """Fixed-address, fixed-capacity batch buffers in the ragged-compact layout (valid node rows only).
        augmented_features: see kgcn_amd.ragged.StaticRaggedBatch (models.wants_augmented_features(model, F) says when it pays)."""
        from .ragged import StaticRaggedBatch
        return StaticRaggedBatch(self, batch_size, capacity, augmented_features=augmented_features)

    def ragged_batch(self, batch_idx, batch_size=None):
        """One mini-batch assembled on the device directly in the ragged-compact layout (exact capacity R + 1)."""
        from .ragged import StaticRaggedBatch
        batch_idx = np.asarray(batch_idx, np.int64).reshape(-1)
        T = batch_idx.shape[0] if batch_size is None else int(batch_size)
        R = int(self.sizes[batch_idx].sum())
        return StaticRaggedBatch(self, T, capacity=(R + 1 + 3) // 4 * 4).load(batch_idx).ragged


class StaticBatch:
    """A mini-batch at FIXED device addresses: adjacency containers (A, A^T and -- for graphs of at most
    32 nodes -- their row-padded copies, pre-linked so that transpose()/padded4() return them) and the
    feature tensor.  load(batch_idx) refills them on the device; the kernels captured in a hipGraph keep
    reading the same pointers."""

    def __init__(self, dataset, batch_size, fused=True):
        import torch
        self.dataset = dataset
        self.batch_size = T = int(batch_size)
        self._sources, chans = [], []
        for src in dataset.channels:
            srcs = [src, src.transpose()]
            padded = fused and src.rows <= BatchedCSR.PAD_COL and src.cols <= BatchedCSR.PAD_COL
            if padded:
                srcs += [src.padded4(), src.transpose().padded4()]
            st = [BatchedCSR.static_like(x, T) for x in srcs]
            st[0]._t, st[1]._t = st[1], st[0]
            if padded:
                st[0]._p4, st[1]._p4 = st[2], st[3]
            self._sources.append(list(zip(srcs, st)))
            chans.append(st[0])
        self.adjacency = BatchedAdjacency(chans)
        f = dataset.features
        self.features = None if f is None else f.new_zeros((T,) + tuple(f.shape[1:]))
        self._pruned = False
        self._sel_dev = torch.zeros(T, dtype=torch.int32, device=dataset.channels[0].rowptr.device)
        self._tables, self._features_as_table, self._ring, self._asm_ws = [], True, None, None

    def reset_usage(self):
        """Forget which containers a kernel has received so far (their descriptors are rebuilt on demand): called before the
        warm-up of a capture, so that prune_unused() keeps exactly the containers the CAPTURED step reads -- a model whose
        first call took another route (Keras-style build, layer by layer) must not keep refilling that route's containers."""
        for pairs in self._sources:
            for _, st in pairs:
                st._desc = None
        self.adjacency._desc_arr = None
        self.adjacency._desc_arr_t = None
        self._pruned = False

    def prune_unused(self):
        """After the consumer (e.g. a captured hipGraph) has run once: refill only the containers whose
        descriptor a kernel actually received -- a fused-kernel model reads the two row-padded containers,
        an unfused one A and A^T, never all four."""
        self._pruned = True

    def add_table(self, table):
        """Register a per-graph device table [G, ...] (labels, masks, true sizes as floats ...): returns the static
        [batch_size, ...] float32 buffer that assemble() fills with the selected graphs' rows (zeros for the dummy graphs of
        a short batch) in the same launch as the feature rows."""
        return _add_table(self, table)

    def stage(self, batch_idx):
        """Host half of load(): validate the indices and send them to the device -- ONE asynchronous copy out of a pinned
        staging ring (a pageable source makes every upload wait for the stream to drain, which serialises a pipelined
        training loop: 0.54 instead of 0.32 ms per step at 4,096 graphs of example_jbl/synthetic.jbl)."""
        batch_idx = np.asarray(batch_idx, np.int64).reshape(-1)
        T, nb = self.batch_size, batch_idx.shape[0]
        if nb > T or (nb and (batch_idx.max() >= self.dataset.num_graphs or batch_idx.min() < 0)):
            raise ValueError("batch indices out of range")
        _stage_selection(self, batch_idx, T)
        return self

    def assemble(self):
        """Device half of load(): every container a kernel reads (all four before prune_unused()), the feature rows and the
        registered tables in two launches (kgcn_batch_assemble) -- capturable: GraphedTrainStep(capture_assembly=True) makes
        it the head of the step's hipGraph, and a step is stage() + replay()."""
        import ctypes
        plan = _lib.AssemblePlan()
        n = 0
        keep = []
        for pairs in self._sources:
            for src, st in pairs:
                if self._pruned and st._desc is None:
                    continue
                if n == _lib.ASSEMBLE_MAX_CSR:
                    self._assemble_flush(plan, n, 0)
                    plan, n = _lib.AssemblePlan(), 0
                d = src.desc()
                keep.append(d)
                plan.src[n] = ctypes.pointer(d)
                plan.dst_rowptr[n] = st.rowptr.data_ptr()
                plan.dst_cv[n] = st.cv.data_ptr() if st.nnz else 0
                plan.dst_cv_capacity[n] = st.nnz
                plan.dst_slots[n] = st.slots.data_ptr() if st.slots is not None else 0
                plan.dst_graph_ptr[n] = st._gptr_buf.data_ptr()
                st._graph_counts = None
                n += 1
        k = _fill_tables(self, plan)
        self._assemble_flush(plan, n, k)
        return self

    def _assemble_flush(self, plan, n, k):
        plan.num_csr, plan.num_tables = n, k
        T = self.batch_size
        if self._asm_ws is None:
            import torch
            wsb = _lib.lib.kgcn_batch_assemble_workspace_bytes(T)
            self._asm_ws = torch.empty(max(wsb, 4) // 4, dtype=torch.int32, device=self._sel_dev.device)
        _lib.check(_lib.lib.kgcn_batch_assemble(plan, self._sel_dev.data_ptr(), T, self._asm_ws.data_ptr(),
                                                self._asm_ws.numel() * 4, _lib.current_stream()), "kgcn_batch_assemble")

    def load(self, batch_idx):
        """stage(batch_idx) + assemble()."""
        return self.stage(batch_idx).assemble()


def _add_table(sb, table):
    import torch
    if table.dtype not in (torch.float32, torch.int32) or not table.is_cuda or table.shape[0] != sb.dataset.num_graphs:
        raise ValueError("a table is a float32 or int32 device tensor with one leading row per dataset graph")
    if len(sb._tables) + (1 if getattr(sb, "_features_as_table", False) else 0) >= _lib.ASSEMBLE_MAX_TABLES:
        raise ValueError("at most %d tables per batch" % _lib.ASSEMBLE_MAX_TABLES)
    table = table.contiguous()
    out = table.new_zeros((sb.batch_size,) + tuple(table.shape[1:]))
    sb._tables.append((table, out))               # rows move as 4-byte words: the dtype does not matter to the kernel
    return out


def _fill_tables(sb, plan, with_features=True):
    k = 0
    rows = []
    if with_features and getattr(sb, "_features_as_table", False) and sb.features is not None:
        rows.append((sb.dataset.features, sb.features))
    rows += sb._tables
    for table, out in rows:
        plan.table[k] = table.data_ptr()
        plan.table_out[k] = out.data_ptr()
        plan.row_floats[k] = table[0].numel() if table.shape[0] else 0
        k += 1
    return k


def _stage_selection(sb, batch_idx, T):
    """sel (int32, -1 = dummy graph) -> sb._sel_dev through a ring of pinned buffers, each guarded by the event of its last copy."""
    import torch
    if sb._ring is None:
        sb._ring = [(torch.empty(T, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        sb._ring_pos = 0
    buf, ev = sb._ring[sb._ring_pos]
    sb._ring_pos = (sb._ring_pos + 1) % len(sb._ring)
    ev.synchronize()                              # the copy issued four stagings ago (a no-op unless the host runs far ahead)
    host = buf.numpy()
    nb = batch_idx.shape[0]
    host[:nb] = batch_idx
    host[nb:] = -1
    sb._sel_dev.copy_(buf, non_blocking=True)
    ev.record()


# -------------------------------------------------------------------------------------------------
# block-diagonal batch (kgcn-sparse path, BASELINE config 3)
# -------------------------------------------------------------------------------------------------
class BlockDiagonalBatch:
    """What construct_batched_adjacency_and_feature_matrices returns, device resident:
    adjacency = BatchedAdjacency of C channels with ONE [sumN x sumN] graph each, features
    torch [sumN, input_dim], plus the molecule -> node-range indicator used by the read-out
    (example_model/sparse.py:83-94)."""

    def __init__(self, adjacency, features, sizes, segments):
        self.adjacency = adjacency
        self.features = features
        self.sizes = sizes
        self.segments = segments        # BatchedCSR [1 graph, B rows, sumN cols] of ones
        self._segments_adj = None

    def segments_adjacency(self):
        """The indicator as a one-channel BatchedAdjacency (what ops.bconv takes: aggregation with an activation epilogue)."""
        if self._segments_adj is None:
            self._segments_adj = BatchedAdjacency([self.segments])
        return self._segments_adj


def block_diagonal_batch(size, adj_row, adj_column, adj_values, adj_elem_len, adj_degrees, feature_row,
                         feature_column, feature_values, feature_elem_len, input_dim, max_degree=5,
                         normalize=True, split_adj=False, device="cuda"):
    """kgcn/data_util.py:698-845 without the per-molecule tf.scan (which builds a [B, nnz_total]
    padded index tensor): molecule offsets are ONE np.repeat over the concatenated entries.
    Same argument list and channel semantics as the reference function (normalise: values /
    sqrt(colsum)[col] / sqrt(colsum)[row] in float32; split: channels by clipped per-entry degree
    1..max_degree + identity channel)."""
    import torch
    size = np.asarray(size, np.int64).reshape(-1)
    total = int(size.sum())
    nmol = size.shape[0]
    offset = np.zeros(nmol, np.int64)
    np.cumsum(size[:-1], out=offset[1:])
    elem = np.asarray(adj_elem_len, np.int64).reshape(-1)
    eoff = np.repeat(offset, elem)
    drow = np.asarray(adj_row, np.int64) + eoff
    dcol = np.asarray(adj_column, np.int64) + eoff
    vals = np.asarray(adj_values, np.float32)
    zeros = lambda n: np.zeros(n, np.int64)
    if normalize:
        deg = np.zeros(total, np.float32)
        np.add.at(deg, dcol, vals)
        sq = np.sqrt(deg).astype(np.float32)
        v = ((vals / sq[dcol]).astype(np.float32) / sq[drow]).astype(np.float32)
        chans = [(drow, dcol, v)]
    elif split_adj:
        d = np.clip(np.asarray(adj_degrees, np.int64), 0, max_degree)
        chans = [(drow[d == k], dcol[d == k], vals[d == k]) for k in range(1, max_degree + 1)]
        eye = np.arange(total, dtype=np.int64)
        chans.append((eye, eye, np.ones(total, np.float32)))
    else:
        chans = [(drow, dcol, vals)]
    adjacency = BatchedAdjacency([BatchedCSR.from_arrays(zeros(r.shape[0]), r, c, v, 1, total, total, device=device)
                                  for r, c, v in chans])
    felem = np.asarray(feature_elem_len, np.int64).reshape(-1)
    net = np.zeros((total, int(input_dim)), np.float32)
    net[np.asarray(feature_row, np.int64) + np.repeat(offset, felem), np.asarray(feature_column, np.int64)] = \
        np.asarray(feature_values, np.float32)
    node = np.arange(total, dtype=np.int64)
    segments = BatchedCSR.from_arrays(zeros(total), np.repeat(np.arange(nmol, dtype=np.int64), size), node,
                                      np.ones(total, np.float32), 1, nmol, total, device=device)
    return BlockDiagonalBatch(adjacency, torch.from_numpy(net).to(device), size, segments)
