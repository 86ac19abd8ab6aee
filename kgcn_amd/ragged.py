"""Ragged-compact batches: the layer stack on the VALID node rows only (csrc/ragged.hip).

The reference pads every graph of a batch to max_node_num rows (kgcn/data_util.py:30-37, kgcn/feed.py:127-133) and
hands the true sizes to the layers as `enabled_node_nums`; its ragged layers gather the valid rows, compute on the
stacked [sum n_b, D] matrix and pad back (GraphDense kgcn/layers.py:243-254, GraphBatchNormalization :196-210), every
other layer computes on all padded rows.  A RaggedBatch IS that stacked matrix, built once per batch, and every layer
of the path runs on it unchanged because it is an ordinary batch of ONE graph with a block-diagonal adjacency:

    features   [1, capacity, F]     rows [graph_ptr[b], graph_ptr[b+1]) = the n_b valid rows of graph b; rows >= R are
                                    "padding representatives" (zero features, no adjacency entries)
    adjacency  BatchedAdjacency     C channels, num_graphs = 1, [capacity x capacity] block-diagonal CSR (+ A^T)
    row_count  int32 [1] (device)   = R; what GraphBatchNormalization / ragged GraphDense take as enabled_node_nums

Why the results are those of the padded formulation: padded rows of the reference's layout never feed a valid row
(the adjacency has no entry there), and all of them carry the SAME value at every layer -- GraphConv gives 0 (empty
adjacency row), an activation act(0), an un-ragged GraphDense the constant row act(prev K + bias), a ragged layer 0.
A padding representative row goes through exactly these steps, so row capacity-1 holds that value, GraphGather adds it
(N - n_b) times (quirk Q4: the reference sums the padded rows too), and the gradient of those padded rows reaches the
kernels / biases through the same row.  Requirement: the padded rows of the input features are zero (feed.py pads with
zeros) and no adjacency entry touches a node >= n_b (checked on the device when the batch is built).
Not defined on this layout: GraphMaxPooling (its implicit-zero rule counts the padded columns) and
GraphBatchNormalization in training phase WITHOUT enabled_node_nums (its statistics would include the padded rows).
"""
import os

import numpy as np

from . import _lib
from .batched_csr import BatchedAdjacency, BatchedCSR, as_batched_adjacency


class RaggedBatch:
    def __init__(self, adjacency, features, graph_ptr, num_graphs, n_nodes, capacity, rows=None):
        self.adjacency = adjacency            # BatchedAdjacency, 1 graph of capacity x capacity
        self.features = features              # torch [1, capacity, F] or None
        self.graph_ptr = graph_ptr            # torch int32 [B + 1] (device)
        self.num_graphs = int(num_graphs)     # B
        self.n_nodes = int(n_nodes)           # N = max_node_num of the padded formulation
        self.capacity = int(capacity)
        self.rows = rows                      # host int R when known (None after a device-only refill)
        self.pad_row = self.capacity - 1

    @property
    def row_count(self):
        """int32 [1] device view: R (fixed address -- what a captured hipGraph reads)."""
        return self.graph_ptr[self.num_graphs:self.num_graphs + 1]

    @property
    def device(self):
        return self.graph_ptr.device

    def gather(self, x):
        from . import ops
        return ops.ragged_gather(x, self)

    def expand(self, x, fill="pad"):
        """[1, capacity, D] (or [capacity, D]) -> the padded layout [B, N, D]; padded rows get the padding representative
        row (fill='pad': what the padded formulation holds there) or zeros (fill='zero')."""
        import torch
        x2 = x.reshape(self.capacity, -1).contiguous()
        d = x2.shape[1]
        out = torch.empty((self.num_graphs, self.n_nodes, d), device=x2.device, dtype=torch.float32)
        _lib.check(_lib.lib.kgcn_ragged_expand_rows_f32(_lib.ptr(x2), self.num_graphs, self.n_nodes, d,
                                                        _lib.ptr(self.graph_ptr), self.pad_row if fill == "pad" else -1,
                                                        _lib.ptr(out), _lib.current_stream()), "kgcn_ragged_expand_rows_f32")
        return out

    def compact_rows(self, padded, out=None):
        """[B, N, D] padded tensor -> [1, capacity, D] with this batch's row map (zeros on the rows >= R)."""
        import torch
        padded = padded.contiguous()
        B, N, d = padded.shape
        if (B, N) != (self.num_graphs, self.n_nodes):
            raise ValueError("tensor %s does not match the batch (%d graphs x %d nodes)" % (tuple(padded.shape), self.num_graphs,
                                                                                          self.n_nodes))
        if out is None:
            out = torch.empty((1, self.capacity, d), device=padded.device, dtype=torch.float32)
        _lib.check(_lib.lib.kgcn_ragged_compact_rows_f32(_lib.ptr(padded), None, B, N, d, _lib.ptr(self.graph_ptr),
                                                         self.capacity, _lib.ptr(out), _lib.current_stream()),
                   "kgcn_ragged_compact_rows_f32")
        return out


def _plan(src, sizes_dev, sel_dev, B, graph_ptr, entry_ptr, ws):
    _lib.check(_lib.lib.kgcn_ragged_plan(src.desc(), _lib.ptr(sizes_dev), _lib.ptr(sel_dev), B, _lib.ptr(graph_ptr),
                                         _lib.ptr(entry_ptr), _lib.ptr(ws), ws.numel() * 4, _lib.current_stream()),
               "kgcn_ragged_plan")


def _compact_csr(src, sel_dev, B, graph_ptr, entry_ptr, capacity, rowptr, cv, status):
    _lib.check(_lib.lib.kgcn_ragged_compact_csr(src.desc(), _lib.ptr(sel_dev), B, _lib.ptr(graph_ptr), _lib.ptr(entry_ptr),
                                                capacity, _lib.ptr(rowptr), cv.data_ptr() if cv.shape[0] else 0,
                                                cv.shape[0], _lib.ptr(status), _lib.current_stream()),
               "kgcn_ragged_compact_csr")


def _container(rowptr, cv, capacity, block_ptr=None, n_nodes=0):
    c = BatchedCSR(rowptr, cv, 1, capacity, capacity, max(int(cv.shape[0]), 1))
    if block_ptr is not None and os.environ.get("KGCN_SPMM_BLOCKS") != "0":          # (development A/B: the row-chunk kernel)
        # the row blocks of whole molecules: the aggregation kernels stage a block's rows in LDS once (csrc/spmm.hip, spmm_block_kernel)
        c.block_ptr, c.block_rows_max = block_ptr, _lib.lib.kgcn_ragged_block_rows() + max(int(n_nodes), 1) - 1
    return c


def _new_block_ptr(capacity, device):
    import torch
    return torch.zeros(_lib.lib.kgcn_ragged_num_blocks(int(capacity)) + 1, dtype=torch.int32, device=device)


def _blocks(graph_ptr, B, capacity, block_ptr):
    _lib.check(_lib.lib.kgcn_ragged_blocks(_lib.ptr(graph_ptr), B, capacity, _lib.ptr(block_ptr), _lib.current_stream()),
               "kgcn_ragged_blocks")


def default_capacity(sizes, batch_size, n_nodes):
    """A row capacity that a batch of `batch_size` graphs drawn INDEPENDENTLY AND UNIFORMLY from a dataset with these sizes exceeds
    with negligible probability (mean + 6 sigma of the sum: 1e-9 per batch), rounded up to a multiple of 64, never above the padded
    row count + 1.  The assumption is the sampling: batches bucketed or sorted by size exceed it routinely, and
    StaticRaggedBatch.stage() then raises ValueError (before anything is written) -- such callers pass
    capacity=batch_size * n_nodes + 1 (the padded bound, always sufficient) to static_ragged_batch()."""
    sizes = np.asarray(sizes, np.float64)
    if sizes.size == 0:
        return 64
    cap = batch_size * sizes.mean() + 6.0 * np.sqrt(batch_size) * sizes.std() + 1
    cap = int(-(-cap // 64) * 64)
    return int(min(cap, -(-(batch_size * n_nodes + 1) // 64) * 64))


def compact(features, adj, enabled_node_nums, capacity=None, check=True):
    """Padded batch (features [B, N, F] or None, adjacency adjs[b][ch] / BatchedAdjacency, true sizes enabled_node_nums
    [B]) -> RaggedBatch.  capacity None: R + 1 rounded up to a multiple of 4 (one host read of R when the sizes live on
    the device).  check: read back the device-side validity count (entries touching nodes >= n_b) and raise."""
    import torch
    dev = features.device if features is not None else None
    if isinstance(adj, RaggedBatch):
        return adj
    n_hint = None if features is None else int(features.shape[1])
    a = as_batched_adjacency(adj, n_nodes=n_hint, device=dev if dev is not None else "cuda")
    B, N = a.num_graphs, a.n_nodes
    dev = a.channels[0].rowptr.device
    if torch.is_tensor(enabled_node_nums):
        sizes_dev = enabled_node_nums.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        R = None
    else:
        sz = np.asarray(enabled_node_nums, np.int64).reshape(-1)
        sizes_dev = torch.from_numpy(sz.astype(np.int32)).to(dev)
        R = int(np.clip(sz, 0, N).sum())
    if sizes_dev.numel() != B:
        raise ValueError("enabled_node_nums has %d entries for a batch of %d graphs" % (sizes_dev.numel(), B))
    if a.values is not None:
        # differentiable adjacency values (integrated gradients over `adjs`, kgcn/visualization.py:207-210): the compact container
        # would bake the stored values in and d values would silently be dropped -- that path stays on the padded layout
        raise ValueError("the batch carries differentiable adjacency values: the ragged-compact layout has no d values path "
                         "(build the model with ragged=False for integrated gradients over the adjacency)")
    if R is None and (capacity is not None or not check):
        # an explicit capacity must be checked BEFORE the device writes rows [0, R]: ragged_csr_kernel / ragged_rows_kernel
        # write dst[r0 + r] unbounded (one host read of R)
        R = int(sizes_dev.clamp(0, N).sum().item())
    if capacity is None:
        if R is None:
            R = int(sizes_dev.clamp(0, N).sum().item())
        capacity = (R + 1 + 3) // 4 * 4
    elif R + 1 > capacity:
        raise ValueError("batch holds %d valid rows, capacity %d needs one more for the padding representative" % (R, capacity))
    i32 = dict(device=dev, dtype=torch.int32)
    graph_ptr = torch.empty(B + 1, **i32)
    ws = torch.empty(max(_lib.lib.kgcn_ragged_workspace_bytes(B), 8) // 4, **i32)
    status = torch.zeros(1, **i32)
    block_ptr = _new_block_ptr(capacity, dev)
    chans = []
    for ic, ch in enumerate(a.channels):
        entry_ptr = torch.empty(B + 1, **i32)
        _plan(ch, sizes_dev, None, B, graph_ptr, entry_ptr, ws)
        if ic == 0:
            _blocks(graph_ptr, B, capacity, block_ptr)
        pair = []
        for src in (ch, ch.transpose()):
            rowptr = torch.empty(capacity + 1, **i32)
            cv = torch.empty((src.nnz, 2), **i32)
            _compact_csr(src, None, B, graph_ptr, entry_ptr, capacity, rowptr, cv, status)
            pair.append(_container(rowptr, cv, capacity, block_ptr, N))
        pair[0]._t, pair[1]._t = pair[1], pair[0]
        chans.append(pair[0])
    feat = None
    if features is not None:
        f = features.contiguous()
        if f.dtype != torch.float32 or tuple(f.shape[:2]) != (B, N):
            raise ValueError("features must be float32 [%d, %d, F], got %s %s" % (B, N, f.dtype, tuple(f.shape)))
    if check:
        got = torch.cat([status, graph_ptr[B:B + 1]]).tolist()
        if got[0]:
            raise ValueError("%d adjacency entries touch nodes beyond enabled_node_nums (or do not fit): the batch has no "
                             "ragged-compact form" % got[0])
        if got[1] + 1 > capacity:
            raise ValueError("batch holds %d valid rows, capacity %d needs one more" % (got[1], capacity))
        R = got[1]
    rb = RaggedBatch(BatchedAdjacency(chans), None, graph_ptr, B, N, capacity, rows=R)
    if features is not None:
        from . import ops
        rb.features = ops.ragged_compact_rows(f, rb)        # differentiable: d features reaches the padded tensor
    return rb


def enter(enabled, features, adjs, enabled_node_nums):
    """What a model's forward does first.  Returns (features, adjs, enabled_node_nums, rb): the compact tensors and the
    RaggedBatch when the ragged-compact route applies (`adjs` already is one, or the model asked for it and the true
    sizes are given), the arguments unchanged and rb = None otherwise."""
    if isinstance(adjs, RaggedBatch):
        rb = adjs
    elif enabled and enabled_node_nums is not None and getattr(adjs, "values", None) is None:
        # (differentiable adjacency values -- integrated gradients over `adjs` -- stay on the padded layout: see compact())
        rb = compact(features, adjs, enabled_node_nums)
    else:
        return features, adjs, enabled_node_nums, None
    feats = rb.features
    if features is not None and features.dim() == 3 and tuple(features.shape[:2]) == (1, rb.capacity):
        feats = features                      # the caller's own compact tensor (e.g. requires_grad inputs)
    return feats, rb, rb.row_count, rb


class StaticRaggedBatch:
    """A ragged-compact mini-batch at FIXED device addresses and a FIXED row capacity, refilled on the device from a
    kgcn_amd.data_util.DeviceGraphDataset that knows its graphs' true sizes: load(batch_idx) runs the plan, the two
    container copies per channel and the feature-row copy (6 launches for one channel) -- the kernels captured in a
    hipGraph (kgcn_amd.train.GraphedTrainStep) keep reading the same pointers whatever R the new batch has.
    .features / .adjacency are what the model takes (adjacency = the RaggedBatch)."""

    def __init__(self, dataset, batch_size, capacity=None, augmented_features=False):
        """augmented_features: the feature rows are assembled as [x | 1 | 0 ...] rows of (F + 1 rounded up to 4) floats -- the
        operand of an aggregate-first GraphConv as the model's first layer (layers.GraphConv reads it where it lies instead of
        running ops.augment_ones over the compact rows every step).  `.features` is then a [1, capacity, F] VIEW of that
        buffer (row stride > F): anything else that reads it makes its own contiguous copy."""
        import torch
        if dataset.sizes is None:
            raise ValueError("the dataset was built without `sizes` (true node counts per graph)")
        self.dataset = dataset
        self.batch_size = B = int(batch_size)
        N = dataset.channels[0].rows
        self.capacity = cap = int(capacity) if capacity else default_capacity(dataset.sizes, B, N)
        dev = dataset.channels[0].rowptr.device
        i32 = dict(device=dev, dtype=torch.int32)
        self._sel_dev = torch.zeros(B, **i32)
        self._tables, self._ring, self._asm_ws = [], None, None
        self._graph_ptr = torch.zeros(B + 1, **i32)
        self._ws = torch.empty(max(_lib.lib.kgcn_ragged_workspace_bytes(B), 8) // 4, **i32)
        self.status = torch.zeros(1, **i32)
        self._block_ptr = _new_block_ptr(cap, dev)
        _blocks(self._graph_ptr, B, cap, self._block_ptr)         # an empty batch until load(): padding rows only
        self._chan = []
        chans = []
        for src in dataset.channels:
            worst = B * max(src.max_nnz, 1)
            entry_ptr = torch.zeros(B + 1, **i32)
            pair = [_container(torch.zeros(cap + 1, **i32), torch.zeros((worst, 2), **i32), cap, self._block_ptr, N) for _ in range(2)]
            pair[0]._t, pair[1]._t = pair[1], pair[0]
            for c in pair:
                c._refillable = True
            self._chan.append((src, src.transpose(), entry_ptr, pair))
            chans.append(pair[0])
        f = dataset.features
        self._feat_aug = None
        if f is not None and augmented_features:
            F = int(f.shape[2])
            self._feat_aug = f.new_zeros((cap, (F + 1 + 3) // 4 * 4))
            self._feat_aug[:, F] = 1.0
            feat = self._feat_aug[:, :F].unsqueeze(0)
            feat._kgcn_aug = self._feat_aug          # (a Python attribute of THIS tensor object: the model receives the same object)
        else:
            feat = None if f is None else f.new_zeros((1, cap, f.shape[2]))
        self.ragged = RaggedBatch(BatchedAdjacency(chans), feat, self._graph_ptr, B, N, cap)
        self.features = feat
        self.adjacency = self.ragged

    def add_table(self, table):
        """Register a per-graph device table [G, ...] (labels, masks ...): returns the static [batch_size, ...] buffer that
        assemble() fills with the selected graphs' rows (zeros for dummy graphs), one launch for all tables."""
        from .data_util import _add_table
        return _add_table(self, table)

    def stage(self, batch_idx):
        """Host half of load(): validate, check the row capacity, send the selection to the device (one asynchronous copy out
        of a pinned staging ring)."""
        from .data_util import _stage_selection
        ds = self.dataset
        batch_idx = np.asarray(batch_idx, np.int64).reshape(-1)
        B, nb = self.batch_size, batch_idx.shape[0]
        if nb > B or (nb and (batch_idx.max() >= ds.num_graphs or batch_idx.min() < 0)):
            raise ValueError("batch indices out of range")
        R = int(ds.sizes[batch_idx].sum())
        if R + 1 > self.capacity:
            raise ValueError("batch holds %d valid rows: beyond the static capacity %d" % (R, self.capacity))
        _stage_selection(self, batch_idx, B)
        self.ragged.rows = R
        return self

    def assemble(self):
        """Device half of load(): the plan, the two container copies per channel, the feature rows and the registered tables
        -- capturable (GraphedTrainStep(capture_assembly=True))."""
        from .data_util import _fill_tables
        ds, B = self.dataset, self.batch_size
        for ic, (src, src_t, entry_ptr, pair) in enumerate(self._chan):
            _plan(src, ds.sizes_dev, self._sel_dev, B, self._graph_ptr, entry_ptr, self._ws)
            if ic == 0:
                _blocks(self._graph_ptr, B, self.capacity, self._block_ptr)
            a, t = pair
            _lib.check(_lib.lib.kgcn_ragged_compact_csr_pair(
                src.desc(), src_t.desc(), _lib.ptr(self._sel_dev), B, _lib.ptr(self._graph_ptr), _lib.ptr(entry_ptr), self.capacity,
                _lib.ptr(a.rowptr), a.cv.data_ptr() if a.cv.shape[0] else 0, a.cv.shape[0],
                _lib.ptr(t.rowptr), t.cv.data_ptr() if t.cv.shape[0] else 0, t.cv.shape[0], _lib.ptr(self.status),
                _lib.current_stream()), "kgcn_ragged_compact_csr_pair")       # A and A^T of the channel: one launch
        if self.features is not None:
            f = ds.features
            if self._feat_aug is not None:
                _lib.check(_lib.lib.kgcn_ragged_compact_rows_aug_f32(_lib.ptr(f), _lib.ptr(self._sel_dev), B, f.shape[1], f.shape[2],
                                                                     _lib.ptr(self._graph_ptr), self.capacity,
                                                                     _lib.ptr(self._feat_aug), self._feat_aug.shape[1],
                                                                     _lib.current_stream()), "kgcn_ragged_compact_rows_aug_f32")
            else:
                _lib.check(_lib.lib.kgcn_ragged_compact_rows_f32(_lib.ptr(f), _lib.ptr(self._sel_dev), B, f.shape[1], f.shape[2],
                                                                 _lib.ptr(self._graph_ptr), self.capacity,
                                                                 _lib.ptr(self.features), _lib.current_stream()),
                           "kgcn_ragged_compact_rows_f32")
        if self._tables:
            import torch
            plan = _lib.AssemblePlan()
            plan.num_csr, plan.num_tables = 0, _fill_tables(self, plan, with_features=False)
            if self._asm_ws is None:
                wsb = _lib.lib.kgcn_batch_assemble_workspace_bytes(B)
                self._asm_ws = torch.empty(max(wsb, 4) // 4, dtype=torch.int32, device=self._sel_dev.device)
            _lib.check(_lib.lib.kgcn_batch_assemble(plan, self._sel_dev.data_ptr(), B, self._asm_ws.data_ptr(),
                                                    self._asm_ws.numel() * 4, _lib.current_stream()), "kgcn_batch_assemble")
        return self

    def load(self, batch_idx):
        """stage(batch_idx) + assemble()."""
        return self.stage(batch_idx).assemble()
