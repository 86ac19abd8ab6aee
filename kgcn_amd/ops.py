"""Functional ops of the hot path on torch (ROCm) tensors, each a torch.autograd.Function whose
forward and backward are calls into libkgcn_hip.so through the C ABI.  No arithmetic of the
path is done by torch here: torch provides device memory, the stream and the autograd tape.

Gradients follow the reference's registered gradients:
  Bspmm  kgcn/bspmm_call.py:22-57   d rhs = A^T g ; d values[e] = <g[row_e], rhs[col_e]>
  Bconv  kgcn/bconv_call.py:30-70   the output gradient fans out to every channel
  Bspmdt kgcn/batched_call.py:33-75 d rhs is the stacked [T*K, D] tensor
and TF's MatMul / BiasAdd gradients for the dense part.
"""
import contextlib

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr, require_gpu
from .batched_csr import BatchedAdjacency, BatchedCSR


ACT_CODES = {None: 0, "none": 0, "linear": 0, "sigmoid": 1, "relu": 2, "tanh": 3}
# A/B switch: d pre-activation inside the wide weight-gradient GEMM when the layer input needs no gradient
wgrad_dact_fusion = True
# OPTION (off): weight gradients of the big dense layers on a SIDE HIP stream.  In a backward pass dW / dbias hang off the
# critical chain (d pre-activation -> d inputs -> adjoint aggregation -> the layer below); the chain's aggregation kernels are
# bound by HBM, the weight-gradient GEMMs by the matrix pipe, so overlapping them looked free.  Measured (tools/gpu_round3_i.sh,
# same box, two rounds): cfg4 1.756 / 1.773 ms with the side stream vs 1.747 / 1.751 without, cfg5 3.03 / 3.04 vs 2.98 / 2.99 --
# the step runs at the package power limit (DESIGN.md lesson 4c), so concurrency buys nothing and costs the fork / join.
# The mechanism stays for experiments: the side stream forks from the calling stream when d pre-activation is ready and is
# joined by an autograd end-of-backward callback (inside a hipGraph capture fork and join become graph edges).
side_stream_wgrad = False
side_wgrad_min_rows = 16384
_side_streams = {}
_side_pending = set()


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# ---- deferred second stages of the parameter gradients (kgcn_reduce_defer / kgcn_reduce_flush, include/kgcn_hip.h) -----------
_deferred_keep = []            # workspaces whose partials are still queued


class deferred_reductions:
    """with ops.deferred_reductions(root=loss): loss.backward()  -- the weight-gradient calls inside queue their second stages;
    leaving the block adds all of them in ONE launch.  Parameter gradients are valid only after the block (kgcn_amd.train uses
    it around the backward pass of a training step; plain autograd code never sees deferral).

    A gradient may only wait when NOTHING reads it inside the backward pass.  `root` (the tensor(s) backward() is about to be
    called on) lets the block check that on the autograd graph itself: a parameter whose AccumulateGrad node has more than one
    incoming edge (used twice -- by kgcn ops, by plain torch ops such as an L2 penalty or a tied torch.matmul, or both: autograd
    ADDS the contributions inside the pass), that carries a tensor hook, or whose .grad already exists (accumulated in place
    inside the pass) is excluded from deferral for this backward pass.  Without `root` only uses by kgcn ops are known
    (_count_use): pass it whenever the model may contain anything else."""

    def __init__(self, root=None):
        self.root = root

    def __enter__(self):
        _no_defer_params.clear()
        if self.root is not None:
            _no_defer_params.update(_readers_inside_backward(self.root))
        self.prev = lib.kgcn_reduce_defer(0 if _no_defer_debug() else 1)
        return self

    def __exit__(self, *exc):
        try:
            flush_reductions()
        finally:
            lib.kgcn_reduce_defer(self.prev)
            _no_defer_params.clear()
        return False


_no_defer_params = set()       # id() of parameters whose gradient is read inside the backward pass now running


def _readers_inside_backward(root):
    """id()s of the leaf tensors under `root` whose gradient contribution from a deferring op could be READ before the flush:
    AccumulateGrad nodes with several incoming edges, tensor hooks, an existing .grad."""
    roots = root if isinstance(root, (list, tuple)) else (root,)
    edges, seen, stack = {}, set(), [t.grad_fn for t in roots if t is not None and t.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        for nxt, _ in fn.next_functions:
            if nxt is None:
                continue
            if hasattr(nxt, "variable"):                 # AccumulateGrad
                edges[nxt] = edges.get(nxt, 0) + 1
            elif nxt not in seen:
                stack.append(nxt)
    out = set()
    for acc, n in edges.items():
        v = acc.variable
        if n > 1 or v.grad is not None or getattr(v, "_backward_hooks", None) or getattr(v, "_post_accumulate_grad_hooks", None):
            out.add(id(v))
    return out


def _no_defer_debug():
    import os
    v = os.environ.get("KGCN_NO_DEFER", "")
    if v == "1":
        return True
    if v == "capture":
        return torch.cuda.is_current_stream_capturing()
    if v == "eager":
        return not torch.cuda.is_current_stream_capturing()
    return False


def flush_reductions():
    check(lib.kgcn_reduce_flush(current_stream()), "kgcn_reduce_flush")
    _deferred_keep.clear()


def _keep_until_flush(t):
    if lib.kgcn_reduce_pending() > 0:
        _deferred_keep.append(t)


# How often each parameter tensor entered a deferral-capable op since the step began: a parameter used TWICE receives two
# gradient contributions that autograd adds INSIDE the backward pass -- before the flush -- so only single-use parameters may wait
# (model_multitask.py's ragged execution runs dense2 on the valid rows and once more on the padding representative).
_param_uses = {}


def _count_use(*params):
    for p in params:
        if p is not None:
            _param_uses[id(p)] = _param_uses.get(id(p), 0) + 1


def _single_use(*params):
    return all(p is None or (_param_uses.get(id(p), 0) == 1 and id(p) not in _no_defer_params) for p in params)


class _no_deferral_unless:
    """switches deferral off around one C call whose results are consumed inside the backward pass"""

    def __init__(self, ok):
        self.ok = ok

    def __enter__(self):
        self.prev = None if self.ok else lib.kgcn_reduce_defer(0)

    def __exit__(self, *exc):
        if self.prev is not None:
            lib.kgcn_reduce_defer(self.prev)
        return False


def join_side_streams():
    """Make the current stream wait for every weight gradient still running on a side stream."""
    for key in list(_side_pending):
        torch.cuda.current_stream(torch.device(*key)).wait_stream(_side_streams[key])
    _side_pending.clear()


def _fork_side(device):
    """-> the side stream, forked from the current one; the join is queued for the end of this backward pass."""
    side = _side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    key = (device.type, device.index)
    if key not in _side_pending:
        _side_pending.add(key)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        except RuntimeError:                     # not inside a backward pass (a direct call of .backward of the Function)
            pass
    return side


def act_code(activation):
    """KGCN_ACT_* code of an activation given by name (the ones the reference's models use: tf.sigmoid, tf.nn.relu,
    tf.tanh) or None."""
    try:
        return ACT_CODES[activation]
    except KeyError:
        raise ValueError("unsupported activation %r (None, 'sigmoid', 'relu', 'tanh')" % (activation,)) from None


def _f32c(t, name):
    require_gpu(t, name)
    if t.dtype != torch.float32:
        raise _lib.KgcnHipError("%s must be float32 (the path computes in fp32), got %s" % (name, t.dtype))
    return t.contiguous()


# -------------------------------------------------------------------------------------------------
# raw (non-differentiable) launches
# -------------------------------------------------------------------------------------------------
def bspmm_raw(csr, rhs2d, d, out2d, rhs_ld=None, rhs_gs=None, out_ld=None, out_gs=None, beta=0.0,
              rhs_col=0, out_col=0):
    """out[t] = beta*out[t] + A[t] @ rhs[t] on strided views of 2-D tensors.
    rhs2d: [T*K, rhs_ld] (block t starts at row t*K), columns rhs_col .. rhs_col+d."""
    rhs_ld = rhs2d.stride(0) if rhs_ld is None else rhs_ld
    out_ld = out2d.stride(0) if out_ld is None else out_ld
    rhs_gs = csr.cols * rhs_ld if rhs_gs is None else rhs_gs
    out_gs = csr.rows * out_ld if out_gs is None else out_gs
    check(lib.kgcn_bspmm_f32(csr.desc(), rhs2d.data_ptr() + 4 * rhs_col, rhs_ld, rhs_gs, d,
                             out2d.data_ptr() + 4 * out_col, out_ld, out_gs, float(beta),
                             current_stream()), "kgcn_bspmm_f32")
    return out2d


# -------------------------------------------------------------------------------------------------
# Bspmm / Bspmdt
# -------------------------------------------------------------------------------------------------
class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rhs, values, csr):
        # rhs: [T*K, D] contiguous.  values: None or differentiable [nnz] (CSR order)
        rhs = _f32c(rhs, "rhs")
        if values is not None:
            csr = csr.with_values(_f32c(values, "values"))
        T, M, K = csr.num_graphs, csr.rows, csr.cols
        if rhs.dim() != 2 or rhs.shape[0] != T * K:
            raise _lib.KgcnHipError("rhs must be [T*K, D] = [%d, D], got %s" % (T * K, tuple(rhs.shape)))
        d = rhs.shape[1]
        out = torch.empty((T * M, d), device=rhs.device, dtype=torch.float32)
        bspmm_raw(csr, rhs, d, out)
        ctx.csr = csr
        ctx.has_values = values is not None
        ctx.save_for_backward(rhs)
        return out

    @staticmethod
    def backward(ctx, g):
        (rhs,) = ctx.saved_tensors
        csr = ctx.csr
        g = _f32c(g, "grad")
        d = rhs.shape[1]
        d_rhs = d_val = None
        if ctx.needs_input_grad[0]:
            d_rhs = torch.empty_like(rhs)
            bspmm_raw(csr.transpose(), g, d, d_rhs)
        if ctx.has_values and ctx.needs_input_grad[1]:
            d_val = torch.empty((csr.nnz,), device=rhs.device, dtype=torch.float32)
            check(lib.kgcn_spmm_values_grad_f32(csr.desc(), ptr(g), d, csr.rows * d, ptr(rhs), d,
                                                csr.cols * d, d, ptr(d_val), current_stream()),
                  "kgcn_spmm_values_grad_f32")
        return d_rhs, d_val, None


def bspmm(csr, rhs, values=None):
    """Batched SpMM.  rhs [T, K, D] or [T*K, D] -> same rank ([T, M, D] or [T*M, D])."""
    if rhs.dim() == 3:
        T, K, D = rhs.shape
        return _SpMM.apply(rhs.reshape(T * K, D), values, csr).reshape(T, csr.rows, D)
    return _SpMM.apply(rhs, values, csr)


# -------------------------------------------------------------------------------------------------
# [W_0 | W_1 | ...] and [b_0 | b_1 | ...] of a multi-channel GraphConv: ONE launch (kgcn_copy2d_multi_f32) instead of two torch.cat,
# and in the backward ONE launch that splits the operand's gradient into a fresh contiguous tensor per parameter (the narrowed
# views a cat's backward returns are cloned by AccumulateGrad: 2 C more launches per step)
# -------------------------------------------------------------------------------------------------
def _copy2d(jobs):
    import ctypes
    arr = (_lib.Copy2dJob * len(jobs))(*[_lib.Copy2dJob(*j) for j in jobs])
    check(lib.kgcn_copy2d_multi_f32(ctypes.cast(arr, ctypes.c_void_p), len(jobs), current_stream()), "kgcn_copy2d_multi_f32")


class _CatChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *tensors):
        C = len(tensors) // 2
        ws = [_f32c(t, "kernel") for t in tensors[:C]]
        bs = [_f32c(t, "bias") for t in tensors[C:]]
        din, dout = ws[0].shape
        if any(tuple(w.shape) != (din, dout) for w in ws) or any(b.numel() != dout for b in bs):
            raise _lib.KgcnHipError("cat_channels: the kernels / biases of the channels differ in shape")
        wcat = torch.empty((din, C * dout), device=ws[0].device, dtype=torch.float32)
        bcat = torch.empty((1, C * dout), device=ws[0].device, dtype=torch.float32)
        jobs = [(w.data_ptr(), wcat.data_ptr() + 4 * c * dout, din, dout, dout, C * dout) for c, w in enumerate(ws)]
        jobs += [(b.data_ptr(), bcat.data_ptr() + 4 * c * dout, 1, dout, dout, C * dout) for c, b in enumerate(bs)]
        _copy2d(jobs)
        ctx.shapes = [tuple(t.shape) for t in tensors]
        ctx.dims = (C, din, dout)
        return wcat, bcat

    @staticmethod
    def backward(ctx, gw, gb):
        C, din, dout = ctx.dims
        gw, gb = _f32c(gw, "grad"), _f32c(gb, "grad")
        dws = [torch.empty(ctx.shapes[c], device=gw.device, dtype=torch.float32) for c in range(C)]
        dbs = [torch.empty(ctx.shapes[C + c], device=gw.device, dtype=torch.float32) for c in range(C)]
        jobs = [(gw.data_ptr() + 4 * c * dout, dws[c].data_ptr(), din, dout, C * dout, dout) for c in range(C)]
        jobs += [(gb.data_ptr() + 4 * c * dout, dbs[c].data_ptr(), 1, dout, C * dout, dout) for c in range(C)]
        _copy2d(jobs)
        return tuple(dws) + tuple(dbs)


class _StackChannelRows(torch.autograd.Function):
    """[W_0; b_0; 0; W_1; b_1; 0; ...]  [C dp, dout]: the operand of an aggregate-first multi-channel GraphConv
    ([A_0 X' | A_1 X' | ...] [W_0; b_0; 0; ...], X' = [X | 1 | 0]) in one launch; `pad` = the constant zero rows."""

    @staticmethod
    def forward(ctx, pad, *tensors):
        C = len(tensors) // 2
        ws = [_f32c(t, "kernel") for t in tensors[:C]]
        bs = [_f32c(t, "bias") for t in tensors[C:]]
        din, dout = ws[0].shape
        npad = pad.shape[0]
        dp = din + 1 + npad
        wa = torch.empty((C * dp, dout), device=ws[0].device, dtype=torch.float32)
        jobs = []
        for c in range(C):
            base = wa.data_ptr() + 4 * c * dp * dout
            jobs.append((ws[c].data_ptr(), base, din, dout, dout, dout))
            jobs.append((bs[c].data_ptr(), base + 4 * din * dout, 1, dout, dout, dout))
            if npad:
                jobs.append((pad.data_ptr(), base + 4 * (din + 1) * dout, npad, dout, dout, dout))
        _copy2d(jobs)
        ctx.shapes = [tuple(t.shape) for t in tensors]
        ctx.dims = (C, din, dout, dp)
        return wa

    @staticmethod
    def backward(ctx, gwa):
        C, din, dout, dp = ctx.dims
        gwa = _f32c(gwa, "grad")
        dws = [torch.empty(ctx.shapes[c], device=gwa.device, dtype=torch.float32) for c in range(C)]
        dbs = [torch.empty(ctx.shapes[C + c], device=gwa.device, dtype=torch.float32) for c in range(C)]
        jobs = []
        for c in range(C):
            base = gwa.data_ptr() + 4 * c * dp * dout
            jobs.append((base, dws[c].data_ptr(), din, dout, dout, dout))
            jobs.append((base + 4 * din * dout, dbs[c].data_ptr(), 1, dout, dout, dout))
        _copy2d(jobs)
        return (None,) + tuple(dws) + tuple(dbs)


def stack_channel_rows(ws, bs, pad):
    return _StackChannelRows.apply(pad, *(list(ws) + list(bs)))


class _FanOut(torch.autograd.Function):
    """z[:, c d : (c + 1) d] = A_c @ x for every channel c: ONE launch that reads x once (kgcn_bconv_fanout_f32 on the channels'
    own containers); backward d x = sum_c A_c^T gz_c: the multi-channel aggregation over the transposed containers."""

    @staticmethod
    def forward(ctx, x, adj):
        x = _f32c(x, "inputs")
        C = adj.num_channels
        T, M, K = adj.num_graphs, adj.n_nodes, adj.channels[0].cols
        d = x.shape[1]
        if x.shape[0] != T * K:
            raise _lib.KgcnHipError("inputs must be [T*K, d] = [%d, d], got %s" % (T * K, tuple(x.shape)))
        z = torch.empty((T * M, C * d), device=x.device, dtype=torch.float32)
        check(lib.kgcn_bconv_fanout_f32(adj.desc_array(False), C, ptr(x), None, d, K * d, d, 0, ptr(z), C * d, M * C * d, d,
                                        current_stream()), "kgcn_bconv_fanout_f32")
        ctx.adj, ctx.d = adj, d
        return z

    @staticmethod
    def backward(ctx, gz):
        adj, d = ctx.adj, ctx.d
        if not ctx.needs_input_grad[0]:
            return None, None
        gz = _f32c(gz, "grad")
        C = adj.num_channels
        T, M, K = adj.num_graphs, adj.n_nodes, adj.channels[0].cols
        dx = torch.empty((T * K, d), device=gz.device, dtype=torch.float32)
        check(lib.kgcn_bconv_act_f32(adj.desc_array(True), C, ptr(gz), C * d, M * C * d, d, d, ptr(dx), d, K * d, 0, current_stream()),
              "kgcn_bconv_act_f32")
        return dx, None


def fan_out(adj, x2d):
    """[A_0 x | A_1 x | ...]  [T*M, C d]"""
    return _FanOut.apply(x2d, adj)


def cat_channels(ws, bs):
    """-> ([W_0 | W_1 | ...] [din, C dout], [b_0 | b_1 | ...] [1, C dout]) as differentiable functions of the 2 C parameters."""
    return _CatChannels.apply(*(list(ws) + list(bs)))


# -------------------------------------------------------------------------------------------------
# Bconv: out[t] = sum_c A_c[t] @ rhs_c[t], rhs given as ONE [T*K, C*D] tensor (channel c = columns
# c*D..(c+1)*D) -- exactly what one GEMM with the concatenated kernels produces.
# -------------------------------------------------------------------------------------------------
class _BConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rhs, adj, d, act, *values):
        # values: nothing, or one tensor per channel -- the channel's adjacency values as differentiable inputs
        # (fp32 [nnz_c], CSR order), None for a channel whose stored values are used
        # act: KGCN_ACT_* code applied to the aggregated output inside the kernel (its derivative, expressed in the
        # output, rides in the adjoint aggregation of the backward)
        rhs = _f32c(rhs, "rhs")
        C = adj.num_channels
        T, M, K = adj.num_graphs, adj.n_nodes, adj.channels[0].cols
        if rhs.shape != (T * K, C * d):
            raise _lib.KgcnHipError("rhs must be [T*K, C*D] = [%d, %d], got %s" % (T * K, C * d, tuple(rhs.shape)))
        if values:
            if len(values) != C:
                raise _lib.KgcnHipError("one value tensor (or None) per adjacency channel is required")
            adj = BatchedAdjacency([ch if v is None else ch.with_values(_f32c(v, "values"))
                                    for ch, v in zip(adj.channels, values)])
        out = torch.empty((T * M, d), device=rhs.device, dtype=torch.float32)
        check(lib.kgcn_bconv_act_f32(adj.desc_array(False), C, ptr(rhs), C * d, K * C * d, d, d,
                                     ptr(out), d, M * d, int(act), current_stream()), "kgcn_bconv_act_f32")
        ctx.adj, ctx.d, ctx.act = adj, d, int(act)
        ctx.nvalues = len(values)
        # rhs (the [T*K, C*D] GEMM output) is only read by the d values gradient; without differentiable adjacency values
        # it is NOT retained (one activation-sized tensor per GraphConv layer until backward otherwise)
        ctx.keeps_rhs = bool(values) and any(ctx.needs_input_grad[4:])
        ctx.save_for_backward(*(([rhs] if ctx.keeps_rhs else []) + ([out] if act else [])))
        return out

    @staticmethod
    def backward(ctx, g):
        adj, d, act = ctx.adj, ctx.d, ctx.act
        saved = list(ctx.saved_tensors)
        rhs = saved.pop(0) if ctx.keeps_rhs else None
        aout = saved.pop(0) if act else None
        g = _f32c(g, "grad")
        C = adj.num_channels
        T, M, K = adj.num_graphs, adj.n_nodes, adj.channels[0].cols
        d_rhs = None
        if ctx.needs_input_grad[0]:
            d_rhs = torch.empty((T * K, C * d), device=g.device, dtype=torch.float32)
            # addn_grad: g fans out to every channel -- ONE launch reads it once and writes every channel's column block
            # (d pre-activation = g * act'(out) is formed while the block is staged)
            check(lib.kgcn_bconv_fanout_f32(adj.desc_array(True), C, ptr(g), ptr(aout) if act else None, d, M * d, d, int(act),
                                            ptr(d_rhs), C * d, K * C * d, d, current_stream()), "kgcn_bconv_fanout_f32")
        if act and ctx.nvalues and any(ctx.needs_input_grad[4:]):
            g = activation_backward(aout, g, act)        # d values needs d pre-activation as a tensor
        d_vals = []
        for c in range(ctx.nvalues):
            # kgcn/bconv_call.py:55-67: d values[t][e] = <addn_grad[t][row_e], b[t][col_e]> per graph-channel -- one
            # gather-multiply-reduce launch per channel over the whole batch, the channel's columns of rhs as b
            if not ctx.needs_input_grad[4 + c]:
                d_vals.append(None)
                continue
            ch = adj.channels[c]
            dv = torch.empty((ch.nnz,), device=g.device, dtype=torch.float32)
            check(lib.kgcn_spmm_values_grad_f32(ch.desc(), ptr(g), d, M * d, rhs.data_ptr() + 4 * c * d, C * d,
                                                K * C * d, d, ptr(dv), current_stream()),
                  "kgcn_spmm_values_grad_f32(bconv)")
            d_vals.append(dv)
        return (d_rhs, None, None, None) + tuple(d_vals)


def bconv(adj, rhs_cat, d, values=None, activation=None):
    """values: optional list with one differentiable fp32 [nnz_c] tensor (CSR order) or None per channel.
    activation: None / 'sigmoid' / 'relu' / 'tanh' applied to the aggregated output inside the kernel."""
    act = act_code(activation)
    if values is None:
        return _BConv.apply(rhs_cat, adj, d, act)
    return _BConv.apply(rhs_cat, adj, d, act, *values)


# -------------------------------------------------------------------------------------------------
# stand-alone activation (where no producer kernel can carry it) and the backward of every fused activation
# -------------------------------------------------------------------------------------------------
def activation_backward(act_out, grad, act):
    """grad * act'(.) with the derivative expressed in the activation OUTPUT; returns a new tensor."""
    act_out, grad = _f32c(act_out, "activation output"), _f32c(grad, "grad")
    dpre = torch.empty_like(grad)
    check(lib.kgcn_act_bwd_f32(ptr(act_out), ptr(grad), grad.numel(), int(act), ptr(dpre), current_stream()),
          "kgcn_act_bwd_f32")
    return dpre


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _f32c(x, "inputs")
        y = torch.empty_like(x)
        check(lib.kgcn_act_fwd_f32(ptr(x), x.numel(), int(act), ptr(y), current_stream()), "kgcn_act_fwd_f32")
        ctx.act = int(act)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return activation_backward(y, g, ctx.act), None


def activation(x, name):
    """tf.sigmoid / tf.nn.relu / tf.tanh as one HIP elementwise kernel (forward) and one (backward)."""
    act = act_code(name)
    return x if act == 0 else _Activation.apply(x, act)


# -------------------------------------------------------------------------------------------------
# dense contraction
# -------------------------------------------------------------------------------------------------
def _dense_ws(din, dout, device):
    """Workspace of the wide-layer GEMM (pre-split weight fragments): (bytes, tensor or None)."""
    wsb = lib.kgcn_dense_fwd_workspace_bytes(din, dout)
    if wsb <= 0:
        return 0, None
    return wsb, torch.empty((wsb // 4,), device=device, dtype=torch.float32)


class WeightTables:
    """bf16 fragment tables of the weight operands of wide dense layers (csrc/wtable.hip), split ONCE per training step.

    dense() registers every weight it sees that has a table route (both orientations: W for the forward, W^T for d input).
    refresh() -- called at the start of a step by kgcn_amd.train (train_step / GraphedTrainStep) -- splits all registered
    operands with one launch and stamps them with the current epoch; invalidate() -- called by TFAdam.step, which rewrites the
    weights through raw pointers -- starts a new epoch.  A lookup returns a table only if it was refreshed in this epoch AND
    the weight's torch version counter is the one seen at the refresh (copy_ / load_state_dict bump it); otherwise dense()
    splits on its own as before.  Only weights a dense() call touched since the previous refresh are split again (a second live
    model costs nothing).  Inside a captured hipGraph the refresh is part of the graph, so every replay rebuilds the
    tables from the weights it is about to use.
    WHO MUST CALL invalidate(): anything that writes a registered weight WITHOUT bumping its version counter between refresh() and
    the forward pass of the same step -- writes through `.data` (p.data.copy_, dist.broadcast(p.data)), custom optimisers working on raw
    pointers.  kgcn_amd.train.TFAdam does; torch.optim optimisers and copy_ / load_state_dict on the parameter itself bump the
    counter.  A write that does neither makes dense() multiply with the tables of the OLD weights, with no error."""

    def __init__(self):
        self.epoch = 0
        self.entries = {}          # (data_ptr, shape) -> entry

    class _Entry:
        __slots__ = ("ref", "tables", "epoch", "version", "used", "ref2", "rows")

    def _entry(self, w, create):
        import weakref
        key = (w.data_ptr(), tuple(w.shape))
        e = self.entries.get(key)
        if e is not None and e.ref() is None:
            del self.entries[key]
            e = None
        if e is None and create and isinstance(w, torch.nn.Parameter):
            din, dout = w.shape
            sizes = [int(lib.kgcn_dense_fwd_workspace_bytes(din, dout)), int(lib.kgcn_dense_fwd_workspace_bytes(dout, din))]
            if max(sizes) <= 0:
                return None
            e = WeightTables._Entry()
            e.ref = weakref.ref(w)
            e.tables = [None if b <= 0 else torch.empty((b // 4,), device=w.device, dtype=torch.float32) for b in sizes]
            e.epoch, e.version, e.used = -1, -1, False
            self.entries[key] = e
        if e is not None:
            e.used = True
        return e

    def register(self, w):
        if enabled_weight_tables and w.dim() == 2 and w.is_cuda and w.is_contiguous():
            self._entry(w, True)

    def lookup(self, w, trans):
        """-> (table tensor, bytes) ready for (w, trans) or (None, 0)."""
        if not enabled_weight_tables:
            return None, 0
        e = self._entry(w, False)
        if e is None or e.epoch != self.epoch or e.tables[trans] is None:
            return None, 0
        p = e.ref()
        if p is None or p._version != e.version:
            return None, 0
        return e.tables[trans], e.tables[trans].numel() * 4

    # ---- stacked operands [w; bias; 0] of the aggregate-first GraphConv (layers.py): split straight from the two parameters by a
    # table job with an extra row -- no torch.cat per step, no table split of its own in front of the layer's GEMM
    def _stacked_key(self, w, bias, rows):
        return ("stacked", w.data_ptr(), bias.data_ptr(), int(rows), tuple(w.shape))

    def register_stacked(self, w, bias, rows):
        if not (enabled_weight_tables and isinstance(w, torch.nn.Parameter) and isinstance(bias, torch.nn.Parameter) and w.is_cuda and
                w.is_contiguous() and bias.is_contiguous() and bias.numel() == w.shape[1] and rows > w.shape[0]):
            return
        key = self._stacked_key(w, bias, rows)
        e = self.entries.get(key)
        if e is None or e.ref() is None or e.ref2() is None:
            import weakref
            b = int(lib.kgcn_dense_fwd_workspace_bytes(rows, w.shape[1]))
            if b <= 0:
                return
            e = WeightTables._Entry()
            e.ref, e.ref2 = weakref.ref(w), weakref.ref(bias)
            e.tables = [torch.empty((b // 4,), device=w.device, dtype=torch.float32), None]
            e.epoch, e.version, e.used, e.rows = -1, (-1, -1), False, int(rows)
            self.entries[key] = e
        e.used = True

    def lookup_stacked(self, w, bias, rows):
        """-> (table tensor, bytes) of [w; bias; 0] with `rows` rows, refreshed in this epoch for these parameter versions, or (None, 0)."""
        if not enabled_weight_tables:
            return None, 0
        e = self.entries.get(self._stacked_key(w, bias, rows))
        if e is None or e.epoch != self.epoch:
            return None, 0
        p, q = e.ref(), e.ref2()
        if p is None or q is None or (p._version, q._version) != e.version:
            return None, 0
        e.used = True
        return e.tables[0], e.tables[0].numel() * 4

    def invalidate(self):
        self.epoch += 1

    def refresh(self):
        import ctypes
        _param_uses.clear()                            # a training step begins here (kgcn_amd.train calls refresh() first)
        live, stacked = [], []
        for key, e in list(self.entries.items()):
            p = e.ref()
            if key[0] == "stacked":
                q = e.ref2()
                if p is None or q is None or p.data_ptr() != key[1] or q.data_ptr() != key[2]:
                    del self.entries[key]
                elif e.used:
                    e.used = False
                    stacked.append((p, q, e))
                continue
            if p is None or p.data_ptr() != key[0]:
                del self.entries[key]
                continue
            if not e.used:
                continue                          # no dense() call touched this weight since the last refresh (another model's)
            e.used = False
            live.append((p, e))
        if not (live or stacked) or not enabled_weight_tables:
            return
        jobs = (_lib.WtableJob * (2 * len(live) + len(stacked)))()
        n = 0
        for p, q, e in stacked:
            din, dout = p.shape
            jobs[n] = _lib.WtableJob(p.data_ptr(), dout, 0, e.rows, dout, din, e.tables[0].data_ptr(), q.data_ptr())
            n += 1
        for p, e in live:
            din, dout = p.shape
            for trans, (k, nn) in enumerate(((din, dout), (dout, din))):
                if e.tables[trans] is not None:
                    jobs[n] = _lib.WtableJob(p.data_ptr(), dout, trans, k, nn, 0, e.tables[trans].data_ptr())
                    n += 1
        check(lib.kgcn_wtable_split_multi(ctypes.cast(jobs, ctypes.c_void_p), n, current_stream()), "kgcn_wtable_split_multi")
        for p, e in live:
            e.epoch, e.version = self.epoch, p._version
        for p, q, e in stacked:
            e.epoch, e.version = self.epoch, (p._version, q._version)


enabled_weight_tables = True
weight_tables = WeightTables()


# One-pass backward of the wide layers (kgcn_dense_bwd_f32, csrc/gemmb.hip): dX, dW and dbias from ONE sweep over (grad, act_out,
# x) -- the d pre-activation tensor is never written and read back (six passes over [m, 256] tensors -> four).
dense_bwd_fusion = True


def _dense_bwd_fused_ok(x2d, w, gy, yact, gp=None, gp_ld=0, n_nodes=0, act_is_smooth=False):
    m, din = x2d.shape
    dout = w.shape[1]
    if not (dense_bwd_fusion and lib.kgcn_dense_bwd_supported(m, din, dout)):
        return False
    ts = [t for t in (x2d, gy, yact) if t is not None]
    if not all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts):
        return False
    if gp is None:
        return True
    # (a sigmoid / tanh layer that is read out AND handed on: that kernel form spills registers inside its loop -- two-call route)
    if gy is not None and yact is not None and act_is_smooth:
        return False
    return gp.data_ptr() % 16 == 0 and gp_ld % 4 == 0 and n_nodes >= 8


def _dense_bwd_fused(ctx, x2d, w, gy, yact, act, need_b, gp=None, gp_ld=0, n_nodes=0):
    """-> (dx, dw, db) of y = act(x2d @ w + bias) from one launch (+ the deferrable second stage of dw / db)."""
    m, din = x2d.shape
    dout = w.shape[1]
    dx = torch.empty_like(x2d)
    dw = torch.empty_like(w)
    db = torch.empty((dout,), device=x2d.device, dtype=torch.float32) if need_b else None
    tab, tb = weight_tables.lookup(w, 1)
    ready = 1
    if tab is None:
        tb, tab = _dense_ws(dout, din, x2d.device)
        ready = 0
    with _no_deferral_unless(getattr(ctx, "defer_ok", False) and _single_use(*ctx.defer_ids)):
        wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout)
        wsp = torch.empty((max(wsb, 4) // 4,), device=x2d.device, dtype=torch.float32)
        check(lib.kgcn_dense_bwd_f32(ptr(gy), None if gp is None else gp.data_ptr(), gp_ld, n_nodes, ptr(yact) if act else None,
                                     int(act), dout, ptr(x2d), din, m, din, dout, ptr(w), dout, ptr(dx), din, ptr(dw), ptr(db),
                                     ptr(tab), tb, ready, ptr(wsp), wsb, current_stream()), "kgcn_dense_bwd_f32")
        _keep_until_flush(wsp)
    if db is not None:
        db = db.reshape(ctx.bias_shape)
    return dx, dw, db


class _Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, w, bias, act=0):
        x2d, w = _f32c(x2d, "x"), _f32c(w, "w")
        m, din = x2d.shape
        dout = w.shape[1]
        if w.shape[0] != din:
            raise _lib.KgcnHipError("kernel is %s but inputs have %d features" % (tuple(w.shape), din))
        b = None if bias is None else _f32c(bias, "bias").reshape(-1)
        if b is not None and b.numel() != dout:
            raise _lib.KgcnHipError("bias has %d elements, expected %d" % (b.numel(), dout))
        y = torch.empty((m, dout), device=x2d.device, dtype=torch.float32)
        tab, tb = weight_tables.lookup(w, 0)
        if tab is not None:
            check(lib.kgcn_dense_fwd_tab_f32(ptr(x2d), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout,
                                             dout, int(act), ptr(tab), tb, current_stream()), "kgcn_dense_fwd_tab_f32")
        else:
            weight_tables.register(w)
            wsb, wsp = _dense_ws(din, dout, x2d.device)
            check(lib.kgcn_dense_fwd_ws_f32(ptr(x2d), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout,
                                            dout, int(act), ptr(wsp), wsb, current_stream()), "kgcn_dense_fwd_ws_f32")
        ctx.act = int(act)
        ctx.save_for_backward(x2d, w, y if act else x2d)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        # the second stage of dW / dbias may only wait (ops.deferred_reductions) when NOTHING reads them inside the backward pass:
        # true for leaf parameters, false for a weight that is itself computed (the concatenated kernels of a multi-channel
        # GraphConv: autograd slices its gradient right away; stack_rows' output is sliced into views only: see _StackRows)
        ctx.defer_ok = bool((w.is_leaf or getattr(w, "_kgcn_defer_safe", False)) and (bias is None or bias.is_leaf))
        # who receives this gradient: the operand itself, or -- for stack_rows' [w; bias; pad], a fresh tensor per call -- the
        # PARAMETERS behind it (a GraphConv applied twice shares them: both contributions are added inside the pass)
        under = getattr(w, "_kgcn_defer_params", None)
        ctx.defer_ids = (w, bias) if under is None else tuple(under) + (bias,)
        if under is None:
            _count_use(w)
        _count_use(bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2d, w, yact = ctx.saved_tensors
        gy = _f32c(gy, "grad")
        m, din = x2d.shape
        dout = w.shape[1]
        dx = dw = db = None
        need_b = ctx.bias_shape is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not (side_stream_wgrad and m >= side_wgrad_min_rows) and \
                _dense_bwd_fused_ok(x2d, w, gy, yact if ctx.act else None):
            # wide layer that hands a gradient on: dX, dW and dbias in ONE pass over (gy, y, x)
            dx, dw, db = _dense_bwd_fused(ctx, x2d, w, gy, yact, ctx.act, need_b)
            return dx, dw, db, None
        # first layer of a model (no d inputs), wide layer: d pre-activation is formed while the weight-gradient GEMM stages
        # the gradient rows -- no elementwise pass over the [m, dout] tensor
        fuse_dact = bool(ctx.act) and not ctx.needs_input_grad[0] and wgrad_dact_fusion and \
            bool(lib.kgcn_dense_wgrad_dact_supported(din, dout))
        if ctx.act and ctx.needs_input_grad[0]:
            # d pre-activation is produced by the dX GEMM itself while it stages the gradient rows (wide layers); the
            # weight-gradient GEMM reads it afterwards
            dx = torch.empty_like(x2d)
            dpre = torch.empty_like(gy)
            tab, tb = weight_tables.lookup(w, 1)
            if tab is not None:
                check(lib.kgcn_dense_dx_dact_tab_f32(ptr(gy), ptr(yact), m, dout, dout, ptr(w), dout, din, ptr(dx), din, ctx.act,
                                                     ptr(dpre), ptr(tab), tb, current_stream()), "kgcn_dense_dx_dact_tab_f32")
            else:
                wsb, wsp = _dense_ws(dout, din, gy.device)
                check(lib.kgcn_dense_dx_dact_f32(ptr(gy), ptr(yact), m, dout, dout, ptr(w), dout, din, ptr(dx), din, ctx.act,
                                                 ptr(dpre), ptr(wsp), wsb, current_stream()), "kgcn_dense_dx_dact_f32")
            gy = dpre
        else:
            if ctx.act and not fuse_dact:
                gy = activation_backward(yact, gy, ctx.act)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x2d)
                # dx = gy @ w^T : w [din, dout] used transposed
                tab, tb = weight_tables.lookup(w, 1)
                if tab is not None:
                    check(lib.kgcn_dense_fwd_tab_f32(ptr(gy), m, dout, dout, ptr(w), dout, 1, None, ptr(dx),
                                                     din, din, 0, ptr(tab), tb, current_stream()), "kgcn_dense_fwd_tab_f32(dx)")
                else:
                    wsb, wsp = _dense_ws(dout, din, gy.device)
                    check(lib.kgcn_dense_fwd_ws_f32(ptr(gy), m, dout, dout, ptr(w), dout, 1, None, ptr(dx),
                                                    din, din, 0, ptr(wsp), wsb, current_stream()), "kgcn_dense_fwd_ws_f32(dx)")
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.bias_shape is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            side = None
            if side_stream_wgrad and m >= side_wgrad_min_rows:
                side = _fork_side(gy.device)
                for t in (gy, x2d, yact):
                    t.record_stream(side)        # the caching allocator must not hand their memory out while `side` reads it
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                dw, db = _Dense._wgrad(ctx, x2d, w, gy, yact, m, din, dout, need_w, need_b, fuse_dact)
        return dx, dw, db, None

    @staticmethod
    def _wgrad(ctx, x2d, w, gy, yact, m, din, dout, need_w, need_b, fuse_dact):
        db = None
        with _no_deferral_unless(getattr(ctx, "defer_ok", False) and _single_use(*ctx.defer_ids)):
            wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout)
            wsp = torch.empty((max(wsb, 4) // 4,), device=gy.device, dtype=torch.float32)
            dw = torch.empty_like(w) if need_w else None
            db = torch.empty((dout,), device=gy.device, dtype=torch.float32) if need_b else None
            if fuse_dact:
                check(lib.kgcn_dense_wgrad_dact_f32(ptr(x2d), din, ptr(gy), ptr(yact), dout, ctx.act, m, din, dout, ptr(dw),
                                                    ptr(db), ptr(wsp), wsb, current_stream()), "kgcn_dense_wgrad_dact_f32")
            else:
                check(lib.kgcn_dense_wgrad_f32(ptr(x2d), din, ptr(gy), dout, m, din, dout, ptr(dw),
                                               ptr(db), ptr(wsp), wsb, current_stream()),
                      "kgcn_dense_wgrad_f32")
            _keep_until_flush(wsp)
            if db is not None:
                db = db.reshape(ctx.bias_shape)
        return dw, db


class _StackRows(torch.autograd.Function):
    """[w; bias; pad] along dim 0 (the operand of an aggregate-first GraphConv, layers.py) with a backward that hands out ROW BLOCKS
    of the incoming gradient as views -- nothing reads the gradient's data inside the backward pass, so the second stage of the
    weight-gradient GEMM that produces it may wait for the step's one reduction launch like a leaf parameter's (the output carries
    `_kgcn_defer_safe`; torch.cat's backward does the same slicing, but that is torch's business and not a contract)."""

    @staticmethod
    def forward(ctx, w, bias, pad):
        ctx.rows = (w.shape[0], bias.shape[0])
        return torch.cat([w, bias, pad], dim=0)

    @staticmethod
    def backward(ctx, g):
        n0, n1 = ctx.rows
        return g[:n0], g[n0:n0 + n1], None


def stack_rows(w, bias, pad):
    out = _StackRows.apply(w, bias, pad)
    out._kgcn_defer_safe = bool(w.is_leaf and bias.is_leaf)
    out._kgcn_defer_params = (w, bias)         # _Dense counts / checks THESE: `out` is a new tensor on every call
    _count_use(w, bias)
    return out


class _DenseStacked(torch.autograd.Function):
    """y = act(x2d @ [w; bias; 0]) for an input that needs NO gradient (the first layer's aggregate-first form, layers.py): the operand
    is never assembled -- the forward reads its fragment table (split from w and bias by the step's one table launch), the backward is ONE
    weight-gradient GEMM over x2d whose row blocks ARE d w and d bias (views: nothing reads them inside the pass, so the second stage
    waits for the step's one reduction launch like a leaf parameter's)."""

    @staticmethod
    def forward(ctx, x2d, w, bias, act, tab, tb):
        m, rows = x2d.shape
        dout = w.shape[1]
        y = torch.empty((m, dout), device=x2d.device, dtype=torch.float32)
        check(lib.kgcn_dense_fwd_tab_f32(ptr(x2d), m, rows, rows, ptr(w), dout, 0, None, ptr(y), dout, dout, int(act), ptr(tab), tb,
                                         current_stream()), "kgcn_dense_fwd_tab_f32")
        ctx.act = int(act)
        ctx.save_for_backward(x2d, w, bias, y)
        ctx.defer_ok = bool(w.is_leaf and bias.is_leaf)
        ctx.defer_ids = (w, bias)
        _count_use(w, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2d, w, bias, y = ctx.saved_tensors
        gy = _f32c(gy, "grad")
        m, rows = x2d.shape
        din, dout = w.shape
        fuse_dact = bool(ctx.act) and wgrad_dact_fusion and bool(lib.kgcn_dense_wgrad_dact_supported(rows, dout))
        if ctx.act and not fuse_dact:
            gy = activation_backward(y, gy, ctx.act)
        with _no_deferral_unless(ctx.defer_ok and _single_use(*ctx.defer_ids)):
            wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, rows, dout)
            wsp = torch.empty((max(wsb, 4) // 4,), device=gy.device, dtype=torch.float32)
            dwa = torch.empty((rows, dout), device=gy.device, dtype=torch.float32)
            if fuse_dact:
                check(lib.kgcn_dense_wgrad_dact_f32(ptr(x2d), rows, ptr(gy), ptr(y), dout, ctx.act, m, rows, dout, ptr(dwa), None,
                                                    ptr(wsp), wsb, current_stream()), "kgcn_dense_wgrad_dact_f32")
            else:
                check(lib.kgcn_dense_wgrad_f32(ptr(x2d), rows, ptr(gy), dout, m, rows, dout, ptr(dwa), None, ptr(wsp), wsb,
                                               current_stream()), "kgcn_dense_wgrad_f32")
            _keep_until_flush(wsp)
        return None, dwa[:din], dwa[din:din + 1].reshape(bias.shape), None, None, None


def dense_stacked(x2d, w, bias, activation=None):
    """act(x2d @ [w; bias; 0]) with x2d [m, rows >= din + 1] -- the GEMM of an aggregate-first GraphConv (A [X | 1 | 0]) [W; b; 0].
    Without a concatenation when the operand's table is ready (inside a training step) and x2d needs no gradient; else stack_rows + dense."""
    rows = x2d.shape[1]
    weight_tables.register_stacked(w, bias, rows)
    # (>= 1,024 rows: below that the dense entry point does not take the table route and would read `rows` rows of w itself)
    if not x2d.requires_grad and x2d.dtype == torch.float32 and x2d.is_contiguous() and bias.dim() == 2 and bias.shape[0] == 1 and \
            x2d.shape[0] >= 1024:
        tab, tb = weight_tables.lookup_stacked(w, bias, rows)
        if tab is not None:
            return _DenseStacked.apply(x2d, w, bias, act_code(activation), tab, tb)
    pad = w.new_zeros((rows - w.shape[0] - 1, w.shape[1]))
    return dense(x2d, stack_rows(w, bias, pad), None, activation=activation)


def dense(x2d, w, bias=None, activation=None):
    """y = act(x2d @ w + bias); the activation rides in the GEMM epilogue."""
    return _Dense.apply(x2d, w, bias, act_code(activation))


class _DenseGather(torch.autograd.Function):
    """y = act(x W + b) of a layer whose output is read out by GraphGather -- and also handed on to the next layer when the
    caller uses y (example_model/model_gin.py:45-60): returns (y [T, N, dout], pooled [T, dout]).  Backward: the incoming
    gradient of node row r is d y[r] + d pooled[r / N]; for wide activated layers the broadcast is formed inside the dX GEMM
    (kgcn_dense_dx_dact_gather_f32) instead of being written out by kgcn_graph_gather_bwd(_add)_f32 and read back."""

    @staticmethod
    def forward(ctx, x2d, w, bias, act, T, N, join=None, join_col=0):
        # join: a [T, wide] buffer whose columns join_col .. join_col + dout receive the read-out (the concatenation of several
        # read-outs, model_gin.py:61, without a concatenation pass: see join_columns)
        x2d, w = _f32c(x2d, "x"), _f32c(w, "w")
        m, din = x2d.shape
        dout = w.shape[1]
        if w.shape[0] != din or m != T * N:
            raise _lib.KgcnHipError("kernel %s / %d graphs of %d nodes do not match inputs %s" % (tuple(w.shape), T, N, tuple(x2d.shape)))
        b = None if bias is None else _f32c(bias, "bias").reshape(-1)
        y = torch.empty((m, dout), device=x2d.device, dtype=torch.float32)
        tab, tb = weight_tables.lookup(w, 0)
        if tab is not None:
            check(lib.kgcn_dense_fwd_tab_f32(ptr(x2d), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout,
                                             dout, int(act), ptr(tab), tb, current_stream()), "kgcn_dense_fwd_tab_f32")
        else:
            weight_tables.register(w)
            wsb, wsp = _dense_ws(din, dout, x2d.device)
            check(lib.kgcn_dense_fwd_ws_f32(ptr(x2d), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout,
                                            dout, int(act), ptr(wsp), wsb, current_stream()), "kgcn_dense_fwd_ws_f32")
        if join is None:
            pooled = torch.empty((T, dout), device=x2d.device, dtype=torch.float32)
            check(lib.kgcn_graph_gather_fwd_f32(ptr(y), T, N, dout, ptr(pooled), current_stream()), "kgcn_graph_gather_fwd_f32")
        else:
            if join.dtype != torch.float32 or join.dim() != 2 or join.shape[0] != T or not join.is_contiguous() or \
                    join_col < 0 or join_col + dout > join.shape[1]:
                raise _lib.KgcnHipError("join buffer %s does not take a [%d, %d] read-out at column %d" % (tuple(join.shape), T, dout, join_col))
            pooled = join[:, join_col:join_col + dout]
            check(lib.kgcn_graph_gather_fwd_ld_f32(ptr(y), T, N, dout, pooled.data_ptr(), join.shape[1], current_stream()),
                  "kgcn_graph_gather_fwd_ld_f32")
        ctx.act, ctx.T, ctx.N = int(act), int(T), int(N)
        ctx.save_for_backward(x2d, w, y)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.defer_ok = bool(w.is_leaf and (bias is None or bias.is_leaf))
        ctx.defer_ids = (w, bias)
        _count_use(w, bias)
        ctx.set_materialize_grads(False)
        return y.view(T, N, dout), pooled

    @staticmethod
    def backward(ctx, gy, gp):
        x2d, w, y = ctx.saved_tensors
        m, din = x2d.shape
        dout = w.shape[1]
        T, N = ctx.T, ctx.N
        if gy is None and gp is None:
            return None, None, None, None, None, None, None, None
        gy = None if gy is None else _f32c(gy.reshape(m, dout), "grad")
        gp_ld = dout
        if gp is not None:
            # a column block of a wider gradient (the backward of join_columns / torch.cat) is read where it lies
            strided = gp.dtype == torch.float32 and gp.dim() == 2 and gp.stride(1) == 1 and gp.stride(0) % 4 == 0 and \
                gp.stride(0) >= dout and gp.data_ptr() % 16 == 0
            if strided and ctx.act and ctx.needs_input_grad[0] and lib.kgcn_dense_dx_dact_gather_supported(m, din, dout):
                gp_ld = gp.stride(0)
            else:
                gp = _f32c(gp, "grad")
        dx = dw = db = None
        need_x = ctx.needs_input_grad[0]
        need_b = ctx.bias_shape is not None and ctx.needs_input_grad[2]
        if need_x and ctx.needs_input_grad[1] and (gp is None or (ctx.act and gp.dtype == torch.float32 and gp.dim() == 2 and
                                                                  gp.stride(1) == 1)) and \
                _dense_bwd_fused_ok(x2d, w, gy, y if ctx.act else None, gp, gp_ld if gp is not None else 0, N, ctx.act in (1, 3)):
            # the whole backward in one pass: the read-out's gradient joins (or stands for) the row gradient while the rows are staged
            dx, dw, db = _dense_bwd_fused(ctx, x2d, w, gy, y, ctx.act, need_b, gp, gp_ld if gp is not None else 0, N)
            return dx, dw, db, None, None, None, None, None
        if gp is not None and ctx.act and need_x and lib.kgcn_dense_dx_dact_gather_supported(m, din, dout):
            dx = torch.empty_like(x2d)
            dpre = torch.empty((m, dout), device=x2d.device, dtype=torch.float32)
            tab, tb = weight_tables.lookup(w, 1)
            ready = 1
            if tab is None:
                tb, tab = _dense_ws(dout, din, x2d.device)
                ready = 0
            check(lib.kgcn_dense_dx_dact_gather_f32(ptr(gy), gp.data_ptr(), gp_ld, N, ptr(y), m, dout, dout, ptr(w), dout, din, ptr(dx), din,
                                                    ctx.act, ptr(dpre), ptr(tab), tb, ready, current_stream()),
                  "kgcn_dense_dx_dact_gather_f32")
            g = dpre
        else:
            # incoming gradient as a tensor, then the plain dense backward
            if gp is not None:
                g = torch.empty((m, dout), device=x2d.device, dtype=torch.float32)
                if gy is not None and dout % 4 == 0:
                    check(lib.kgcn_graph_gather_bwd_add_f32(ptr(gp), ptr(gy), T, N, dout, ptr(g), current_stream()),
                          "kgcn_graph_gather_bwd_add_f32")
                else:
                    check(lib.kgcn_graph_gather_bwd_f32(ptr(gp), T, N, dout, ptr(g), current_stream()), "kgcn_graph_gather_bwd_f32")
                    if gy is not None:
                        g = g + gy
            else:
                g = gy
            if ctx.act:
                g = activation_backward(y, g, ctx.act)
            if need_x:
                dx = torch.empty_like(x2d)
                tab, tb = weight_tables.lookup(w, 1)
                if tab is not None:
                    check(lib.kgcn_dense_fwd_tab_f32(ptr(g), m, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0, ptr(tab), tb,
                                                     current_stream()), "kgcn_dense_fwd_tab_f32(dx)")
                else:
                    wsb, wsp = _dense_ws(dout, din, g.device)
                    check(lib.kgcn_dense_fwd_ws_f32(ptr(g), m, dout, dout, ptr(w), dout, 1, None, ptr(dx), din, din, 0, ptr(wsp), wsb,
                                                    current_stream()), "kgcn_dense_fwd_ws_f32(dx)")
        need_w = ctx.needs_input_grad[1]
        need_b = ctx.bias_shape is not None and ctx.needs_input_grad[2]
        if need_w or need_b:
            dw, db = _Dense._wgrad(ctx, x2d, w, g, y, m, din, dout, need_w, need_b, False)
        return dx, dw, db, None, None, None, None, None


def dense_gather(x3d, w, bias=None, activation=None, join=None, join_col=0):
    """GraphDense followed by GraphGather on [T, N, din] inputs: -> (y [T, N, dout], pooled [T, dout]); use y only if the layer
    output is also handed on.  join / join_col: write the read-out into columns join_col.. of the [T, wide] buffer `join`
    (pooled is then that column block; combine the blocks with join_columns)."""
    T, N, din = x3d.shape
    return _DenseGather.apply(x3d.reshape(T * N, din), w, bias, act_code(activation), T, N, join, int(join_col))


class _JoinColumns(torch.autograd.Function):
    """tf.concat(parts, axis=1) of column blocks that were WRITTEN INTO `buf` by their producers (dense_gather(join=buf)): nothing
    is copied, the gradient of part i is the column block i of the gradient -- a strided view its consumer reads in place."""

    @staticmethod
    def forward(ctx, buf, *parts):
        col = 0
        for p in parts:
            if p.data_ptr() != buf.data_ptr() + 4 * col or p.shape[0] != buf.shape[0] or p.stride(0) != buf.stride(0):
                raise _lib.KgcnHipError("join_columns: part at column %d was not produced into the join buffer" % col)
            col += p.shape[1]
        if col != buf.shape[1]:
            raise _lib.KgcnHipError("join_columns: the parts cover %d of %d columns" % (col, buf.shape[1]))
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        out, col = [], 0
        for wd in ctx.widths:
            out.append(g[:, col:col + wd])
            col += wd
        return (None,) + tuple(out)


def join_columns(buf, parts):
    return _JoinColumns.apply(buf, *parts)


# -------------------------------------------------------------------------------------------------
# fused GraphConv (one channel)
# -------------------------------------------------------------------------------------------------
def graphconv_fused_supported(csr, din, dout):
    """Shape test of the fused kernels; they read the row-padded layout (BatchedCSR.padded4()),
    whose per-graph entry count is bounded by max_nnz + 4 * rows."""
    if csr.rows != csr.cols or csr.rows > BatchedCSR.PAD_COL:
        return False
    if csr.row_pad == 0 and csr._p4 is None and csr._make_p4 is None and csr._host is None:
        return False                    # no way to the row-padded copy (ragged-compact / value-substituted containers)
    bound = csr._p4.max_nnz if csr._p4 is not None else csr.max_nnz + 4 * csr.rows
    return bool(lib.kgcn_graphconv_fused_supported(csr.rows, din, dout, bound))


class _GraphConvFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, csr):
        x, w = _f32c(x, "inputs"), _f32c(w, "kernel")
        T, N, din = x.shape
        dout = w.shape[1]
        b = _f32c(bias, "bias").reshape(-1)
        out = torch.empty((T, N, dout), device=x.device, dtype=torch.float32)
        check(lib.kgcn_graphconv_fwd_f32(csr.padded4().desc(), ptr(x), ptr(w), ptr(b), din, dout,
                                         ptr(out), current_stream()), "kgcn_graphconv_fwd_f32")
        ctx.csr = csr
        ctx.bias_shape = tuple(bias.shape)
        ctx.defer_ok = bool(w.is_leaf and bias.is_leaf)
        ctx.defer_ids = (w, bias)
        _count_use(w, bias)
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32c(g, "grad")
        T, N, din = x.shape
        dout = w.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # dW and dbias are the two halves of ONE buffer: the data-parallel exchange all-reduces it where it lies
        # (parallel.GradBucket recognises gradients that already form a contiguous run: no pack / unpack launches)
        dwb = torch.empty((din * dout + dout,), device=x.device, dtype=torch.float32)
        dw, db = dwb[:din * dout].view(din, dout), dwb[din * dout:]
        wsb = lib.kgcn_graphconv_bwd_workspace_bytes(T, din, dout)
        wsp = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        with _no_deferral_unless(ctx.defer_ok and _single_use(*ctx.defer_ids)):
            check(lib.kgcn_graphconv_bwd_f32(ctx.csr.transpose().padded4().desc(), ptr(x), ptr(w),
                                             ptr(g), din, dout, ptr(dx), ptr(dw), ptr(db), ptr(wsp),
                                             wsb, current_stream()), "kgcn_graphconv_bwd_f32")
        _keep_until_flush(wsp)
        return dx, dw, db.reshape(ctx.bias_shape), None


def graphconv_fused(x, w, bias, csr):
    return _GraphConvFused.apply(x, w, bias, csr)


# -------------------------------------------------------------------------------------------------
# GINAggregate
# -------------------------------------------------------------------------------------------------
class _GinAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, adj):
        x = _f32c(x, "inputs")
        T, N, d = x.shape
        if (T, N) != (adj.num_graphs, adj.n_nodes):
            raise _lib.KgcnHipError("inputs %s do not match the adjacency batch (%d graphs x %d nodes)"
                                    % (tuple(x.shape), adj.num_graphs, adj.n_nodes))
        e = None if eps is None else _f32c(eps, "epsilon").reshape(-1)
        out = torch.empty_like(x)
        check(lib.kgcn_gin_aggregate_f32(adj.desc_array(False), adj.num_channels, ptr(x), d, ptr(e),
                                         ptr(out), current_stream()), "kgcn_gin_aggregate_f32")
        ctx.adj = adj
        ctx.save_for_backward(x, e if e is not None else x.new_empty(0))
        ctx.has_eps = e is not None
        ctx.eps_shape = None if eps is None else tuple(eps.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, e = ctx.saved_tensors
        adj = ctx.adj
        g = _f32c(g, "grad")
        T, N, d = x.shape
        dx = deps = None
        want_eps = ctx.has_eps and ctx.needs_input_grad[1]
        if ctx.needs_input_grad[0] or want_eps:
            # one call: dx = sum_c (eps_c g + A_c^T g) and d eps_c = <g, x> (the same inner product for every channel,
            # kgcn/layers.py:469), the inner product accumulated while the gradient tiles are staged for the aggregation
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            one = torch.empty((1,), device=x.device, dtype=torch.float32) if want_eps else None
            wsb = lib.kgcn_gin_aggregate_bwd_workspace_bytes(T, N, d)
            wsp = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
            check(lib.kgcn_gin_aggregate_bwd_f32(adj.desc_array(True), adj.num_channels, ptr(g), d,
                                                 ptr(e) if ctx.has_eps else None, ptr(x), ptr(dx), ptr(one), ptr(wsp), wsb,
                                                 current_stream()), "kgcn_gin_aggregate_bwd_f32")
            if want_eps:
                # one channel (the common case): the kernel's output IS the gradient tensor -- no device-to-device copy per layer
                deps = one.reshape(ctx.eps_shape) if adj.num_channels == 1 else \
                    one.expand(adj.num_channels).reshape(ctx.eps_shape).clone()
        return dx, deps, None


def gin_aggregate(x, eps, adj):
    return _GinAggregate.apply(x, eps, adj)


_GIN_DOT = __import__("os").environ.get("KGCN_GIN_DOT") != "0"           # (development A/B: "0" = the two separate ops)


class _GinDense(torch.autograd.Function):
    """GINAggregate followed by an activated wide GraphDense, for an input that needs NO gradient (the first block of
    example_model/model_gin.py:45-50): y = act((eps x + A x) W + b).  The gradient of the aggregation's output is then only needed for
    d eps = <d out, x> (kgcn/layers.py:469): the layer's dX GEMM accumulates that inner product instead of storing its product
    (kgcn_dense_dx_dact_dot_f32) -- the [rows, din] tensor is neither written nor read back by a dot kernel.  Use gin_dense():
    it falls back to the two separate ops whenever a condition of this form does not hold."""

    @staticmethod
    def forward(ctx, x, eps, adj, w, bias, act):
        x, w = _f32c(x, "inputs"), _f32c(w, "w")
        T, N, din = x.shape
        m, dout = T * N, w.shape[1]
        e = _f32c(eps, "epsilon").reshape(-1)
        agg = torch.empty((m, din), device=x.device, dtype=torch.float32)
        check(lib.kgcn_gin_aggregate_f32(adj.desc_array(False), 1, ptr(x), din, ptr(e), ptr(agg), current_stream()),
              "kgcn_gin_aggregate_f32")
        b = None if bias is None else _f32c(bias, "bias").reshape(-1)
        y = torch.empty((m, dout), device=x.device, dtype=torch.float32)
        tab, tb = weight_tables.lookup(w, 0)
        if tab is not None:
            check(lib.kgcn_dense_fwd_tab_f32(ptr(agg), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, int(act), ptr(tab), tb,
                                             current_stream()), "kgcn_dense_fwd_tab_f32")
        else:
            weight_tables.register(w)
            wsb, wsp = _dense_ws(din, dout, x.device)
            check(lib.kgcn_dense_fwd_ws_f32(ptr(agg), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, int(act), ptr(wsp), wsb,
                                            current_stream()), "kgcn_dense_fwd_ws_f32")
        ctx.act, ctx.eps_shape = int(act), tuple(eps.shape)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.defer_ok = bool(w.is_leaf and (bias is None or bias.is_leaf))
        ctx.defer_ids = (w, bias)
        _count_use(w, bias)
        ctx.save_for_backward(x, agg, w, y)
        return y.view(T, N, dout)

    @staticmethod
    def backward(ctx, gy):
        x, agg, w, y = ctx.saved_tensors
        m, din = agg.shape
        dout = w.shape[1]
        gy = _f32c(gy.reshape(m, dout), "grad")
        deps = torch.empty((1,), device=gy.device, dtype=torch.float32)
        tab, tb = weight_tables.lookup(w, 1)
        ready = 1
        if tab is None:
            tb, tab = _dense_ws(dout, din, gy.device)
            ready = 0
        need_w = ctx.needs_input_grad[3]
        need_b = ctx.bias_shape is not None and ctx.needs_input_grad[4]
        if need_w and _dense_bwd_fused_ok(agg, w, gy, y):
            # ONE pass: d eps, dW and dbias from a single sweep over (gy, y, agg, x) -- no d pre-activation tensor either
            dw = torch.empty_like(w)
            db = torch.empty((dout,), device=gy.device, dtype=torch.float32) if need_b else None
            with _no_deferral_unless(getattr(ctx, "defer_ok", False) and _single_use(*ctx.defer_ids)):
                wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout)
                wsp = torch.empty((max(wsb, 4) // 4,), device=gy.device, dtype=torch.float32)
                check(lib.kgcn_dense_bwd_dot_f32(ptr(gy), ptr(y), ctx.act, dout, ptr(agg), din, m, din, dout, ptr(w), dout, ptr(x), din,
                                                 ptr(dw), ptr(db), ptr(deps), ptr(tab), tb, ready, ptr(wsp), wsb, current_stream()),
                      "kgcn_dense_bwd_dot_f32")
                _keep_until_flush(wsp)
            if db is not None:
                db = db.reshape(ctx.bias_shape)
            return None, (deps.reshape(ctx.eps_shape) if ctx.needs_input_grad[1] else None), None, dw, db, None
        dpre = torch.empty_like(gy)
        wsb = lib.kgcn_dense_dx_dact_dot_workspace_bytes(m, din)
        wsp = torch.empty((max(wsb, 4) // 4,), device=gy.device, dtype=torch.float32)
        check(lib.kgcn_dense_dx_dact_dot_f32(ptr(gy), ptr(y), m, dout, dout, ptr(w), dout, din, ptr(x), din, ctx.act, ptr(dpre),
                                             ptr(tab), tb, ready, ptr(deps), ptr(wsp), wsb, current_stream()),
              "kgcn_dense_dx_dact_dot_f32")
        dw = db = None
        if need_w or need_b:
            dw, db = _Dense._wgrad(ctx, agg, w, dpre, y, m, din, dout, need_w, need_b, False)
        return None, (deps.reshape(ctx.eps_shape) if ctx.needs_input_grad[1] else None), None, dw, db, None


def gin_dense(x, eps, adj, w, bias=None, activation=None):
    """-> act((eps x + A x) @ w + bias), [T, N, dout]; eps: the GINAggregate's epsilon of ONE adjacency channel.  The fused backward
    (no d out tensor: see _GinDense) when x needs no gradient, eps does, the layer is activated and wide enough; else the two ops."""
    T, N, din = x.shape
    act = act_code(activation)
    fused = (not x.requires_grad and eps is not None and eps.requires_grad and eps.numel() == 1 and adj.num_channels == 1 and act and
             torch.is_grad_enabled() and x.dtype == torch.float32 and _GIN_DOT and
             bool(lib.kgcn_dense_dx_dact_dot_supported(T * N, din, w.shape[1])))
    if fused:
        return _GinDense.apply(x, eps, adj, w, bias, act)
    h = gin_aggregate(x, eps, adj)
    return dense(h.reshape(T * N, din), w, bias, activation=activation).reshape(T, N, w.shape[1])


# -------------------------------------------------------------------------------------------------
# GraphMaxPooling
# -------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj):
        x = _f32c(x, "inputs")
        T, N, d = x.shape
        if (T, N) != (adj.num_graphs, adj.channels[0].cols):
            raise _lib.KgcnHipError("inputs %s do not match the adjacency batch" % (tuple(x.shape),))
        out = torch.empty((T, adj.n_nodes, d), device=x.device, dtype=torch.float32)
        for c, ch in enumerate(adj.channels):            # tf.add_n over the channels
            check(lib.kgcn_graph_maxpool_fwd_f32(ch.desc(), ptr(x), d, ptr(out), 0.0 if c == 0 else 1.0,
                                                 current_stream()), "kgcn_graph_maxpool_fwd_f32")
        ctx.adj = adj
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        adj = ctx.adj
        g = _f32c(g, "grad")
        T, N, d = x.shape
        dx = torch.empty_like(x)
        wsb = lib.kgcn_graph_maxpool_bwd_workspace_bytes(T, adj.n_nodes, d)
        wsp = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        for c, ch in enumerate(adj.channels):
            check(lib.kgcn_graph_maxpool_bwd_f32(ch.desc(), ch.transpose().desc(), ptr(x), ptr(g), d, ptr(dx),
                                                 0.0 if c == 0 else 1.0, ptr(wsp), wsb, current_stream()),
                  "kgcn_graph_maxpool_bwd_f32")
        return dx, None


def graph_maxpool(x, adj):
    return _MaxPool.apply(x, adj)


# -------------------------------------------------------------------------------------------------
# GAT
# -------------------------------------------------------------------------------------------------
class _Gat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, *weight_a):
        x = _f32c(x, "inputs")
        T, N, d = x.shape
        if (T, N) != (adj.num_graphs, adj.n_nodes) or adj.channels[0].cols != N:
            raise _lib.KgcnHipError("inputs %s do not match the adjacency batch" % (tuple(x.shape),))
        was = [_f32c(w.reshape(-1), "weight_a") for w in weight_a]
        if len(was) != adj.num_channels or any(w.numel() != 2 * d for w in was):
            raise _lib.KgcnHipError("one weight_a [2*%d, 1] per adjacency channel is required" % d)
        out = torch.empty_like(x)
        wsb = lib.kgcn_gat_workspace_bytes(T, N, d)
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        for c, ch in enumerate(adj.channels):
            check(lib.kgcn_gat_fwd_f32(ch.desc(), ptr(x), d, ptr(was[c]), ptr(out), 0.0 if c == 0 else 1.0, ptr(ws), wsb,
                                       current_stream()), "kgcn_gat_fwd_f32")
        ctx.adj = adj
        ctx.save_for_backward(x, *was)
        return out

    @staticmethod
    def backward(ctx, g):
        x, *was = ctx.saved_tensors
        adj = ctx.adj
        g = _f32c(g, "grad")
        T, N, d = x.shape
        dx = torch.empty_like(x)
        wsb = lib.kgcn_gat_workspace_bytes(T, N, d)
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        dws = []
        for c, ch in enumerate(adj.channels):
            dw = torch.empty(2 * d, device=x.device, dtype=torch.float32)
            check(lib.kgcn_gat_bwd_f32(ch.desc(), ch.transpose().desc(), ptr(x), d, ptr(was[c]), ptr(g), ptr(dx),
                                       0.0 if c == 0 else 1.0, ptr(dw), ptr(ws), wsb, current_stream()),
                  "kgcn_gat_bwd_f32")
            dws.append(dw.reshape(2 * d, 1))
        return (dx, None) + tuple(dws)


def gat(x, adj, weight_a):
    return _Gat.apply(x, adj, *weight_a)


# -------------------------------------------------------------------------------------------------
# decoders: weighted Gram matrix per graph
# -------------------------------------------------------------------------------------------------
class _Gram(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _f32c(x, "inputs")
        T, N, d = x.shape
        wv = None if w is None else _f32c(w.reshape(-1), "kernel")
        out = torch.empty((T, N, N), device=x.device, dtype=torch.float32)
        check(lib.kgcn_gram_fwd_f32(ptr(x), T, N, d, 0 if wv is None else ptr(wv), ptr(out), current_stream()),
              "kgcn_gram_fwd_f32")
        ctx.has_w = wv is not None
        ctx.w_shape = None if w is None else tuple(w.shape)
        ctx.save_for_backward(x, wv if wv is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, g):
        x, wv = ctx.saved_tensors
        g = _f32c(g, "grad")
        T, N, d = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty(d, device=x.device, dtype=torch.float32) if ctx.has_w else None
        wsb = lib.kgcn_gram_workspace_bytes(d)
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        check(lib.kgcn_gram_bwd_f32(ptr(x), T, N, d, ptr(wv) if ctx.has_w else 0, ptr(g), ptr(dx), 0.0,
                                    ptr(dw) if ctx.has_w else 0, ptr(ws), wsb, current_stream()), "kgcn_gram_bwd_f32")
        return dx, (dw.reshape(ctx.w_shape) if ctx.has_w else None)


def gram(x, w=None):
    """out[t] = (x[t] * w) @ x[t]^T  (w: [d] or None)."""
    return _Gram.apply(x, w)


# -------------------------------------------------------------------------------------------------
# GraphGather
# -------------------------------------------------------------------------------------------------
class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x, "inputs")
        B, N, d = x.shape
        out = torch.empty((B, d), device=x.device, dtype=torch.float32)
        check(lib.kgcn_graph_gather_fwd_f32(ptr(x), B, N, d, ptr(out), current_stream()),
              "kgcn_graph_gather_fwd_f32")
        ctx.shape = (B, N, d)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, d = ctx.shape
        g = _f32c(g, "grad")
        dx = torch.empty((B, N, d), device=g.device, dtype=torch.float32)
        check(lib.kgcn_graph_gather_bwd_f32(ptr(g), B, N, d, ptr(dx), current_stream()),
              "kgcn_graph_gather_bwd_f32")
        return dx


def graph_gather(x):
    return _Gather.apply(x)


class _GatherTee(torch.autograd.Function):
    """(x, sum over nodes of x): for a tensor that is read out AND handed to the next layer (model_gin.py:45-60).  Backward:
    d x = d(passed-on x) + broadcast(d pooled) in ONE pass (kgcn_graph_gather_bwd_add_f32) -- separate ops cost a broadcast
    pass plus autograd's accumulation pass over the [B, N, D] tensor."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x, "inputs")
        B, N, d = x.shape
        out = torch.empty((B, d), device=x.device, dtype=torch.float32)
        check(lib.kgcn_graph_gather_fwd_f32(ptr(x), B, N, d, ptr(out), current_stream()), "kgcn_graph_gather_fwd_f32")
        ctx.shape = (B, N, d)
        return x.view_as(x), out

    @staticmethod
    def backward(ctx, gx, gp):
        B, N, d = ctx.shape
        if gp is None:
            return gx
        gp = _f32c(gp, "grad")
        dx = torch.empty((B, N, d), device=gp.device, dtype=torch.float32)
        if gx is None or d % 4 != 0:
            check(lib.kgcn_graph_gather_bwd_f32(ptr(gp), B, N, d, ptr(dx), current_stream()), "kgcn_graph_gather_bwd_f32")
            return dx if gx is None else dx + gx
        gx = _f32c(gx, "grad")
        check(lib.kgcn_graph_gather_bwd_add_f32(ptr(gp), ptr(gx), B, N, d, ptr(dx), current_stream()),
              "kgcn_graph_gather_bwd_add_f32")
        return dx


def graph_gather_tee(x):
    """-> (x, pooled [B, D]); use the returned x downstream."""
    return _GatherTee.apply(x)


class _RaggedGather(torch.autograd.Function):
    """GraphGather on a ragged-compact batch (kgcn_amd.ragged): sum of the valid rows + (N - n_b) x the padding
    representative row (quirk Q4: the reference's reduce_sum runs over the padded rows too)."""

    @staticmethod
    def forward(ctx, x, rb):
        x = _f32c(x, "inputs")
        d = x.shape[-1]
        if x.numel() != rb.capacity * d:
            raise _lib.KgcnHipError("inputs %s do not match the ragged batch (capacity %d rows)" % (tuple(x.shape), rb.capacity))
        out = torch.empty((rb.num_graphs, d), device=x.device, dtype=torch.float32)
        check(lib.kgcn_ragged_gather_fwd_f32(ptr(x), ptr(rb.graph_ptr), rb.num_graphs, rb.n_nodes, d, rb.pad_row, ptr(out),
                                             current_stream()), "kgcn_ragged_gather_fwd_f32")
        ctx.rb, ctx.shape = rb, tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        rb = ctx.rb
        g = _f32c(g, "grad")
        d = ctx.shape[-1]
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        wsb = lib.kgcn_ragged_gather_bwd_workspace_bytes(d)
        ws = torch.empty((wsb // 4,), device=g.device, dtype=torch.float32)
        check(lib.kgcn_ragged_gather_bwd_f32(ptr(g), ptr(rb.graph_ptr), rb.num_graphs, rb.n_nodes, d, rb.pad_row, rb.capacity,
                                             ptr(dx), ptr(ws), wsb, current_stream()), "kgcn_ragged_gather_bwd_f32")
        return dx, None


def ragged_gather(x, rb):
    return _RaggedGather.apply(x, rb)


class _RaggedCompactRows(torch.autograd.Function):
    """Padded [B, N, D] -> the valid rows [1, capacity, D] of a ragged-compact batch; the gradient goes back to the valid
    rows of the padded tensor (zeros on its padded rows)."""

    @staticmethod
    def forward(ctx, padded, rb):
        ctx.rb = rb
        return rb.compact_rows(_f32c(padded, "features"))

    @staticmethod
    def backward(ctx, g):
        return ctx.rb.expand(_f32c(g, "grad"), fill="zero"), None


def ragged_compact_rows(padded, rb):
    return _RaggedCompactRows.apply(padded, rb)


# -------------------------------------------------------------------------------------------------
# cross-layer stack for small graphs (csrc/stack.hip): [GraphConv | GraphDense | BatchNormalization(moving stats)] x k
# (+ GraphGather) in one forward and one backward launch
# -------------------------------------------------------------------------------------------------
# kernels of the cross-layer stack (kgcn_stack_layer.route): 0 automatic, 1 one graph per workgroup trip, 2 64-row tiles (f32 MFMA)
stack_route = 0


def _stack_descriptors(spec, params, buffers):
    """spec: list of (kind, act_code, din, dout, eps); params: flat list, two tensors (or None) per layer (w, b | gamma,
    beta); buffers: per layer (mean, var) or None.  -> (ctypes array of kgcn_stack_layer, keep-alive list)."""
    import ctypes
    arr = (_lib.StackLayer * len(spec))()
    keep = []
    for l, (kind, act, din, dout, eps) in enumerate(spec):
        w, b = params[2 * l], params[2 * l + 1]
        w = _f32c(w.reshape(-1), "stack parameter")
        b = None if b is None else _f32c(b.reshape(-1), "stack parameter")
        mean = var = None
        if kind == 2:
            mean, var = (_f32c(t.reshape(-1), "moving statistics") for t in buffers[l])
        keep += [w, b, mean, var]
        arr[l] = _lib.StackLayer(kind, act, din, dout, w.data_ptr(), 0 if b is None else b.data_ptr(),
                                 0 if mean is None else mean.data_ptr(), 0 if var is None else var.data_ptr(), float(eps),
                                 stack_route if l == 0 else 0)
    return arr, keep


def gcn_stack_supported(csr, spec):
    import ctypes
    if csr.row_pad or csr.rows != csr.cols or csr.rows > 32 or len(spec) > 8:
        return False
    arr = (_lib.StackLayer * len(spec))()
    for l, (kind, act, din, dout, eps) in enumerate(spec):
        arr[l] = _lib.StackLayer(kind, act, din, dout, 1, 1, 1, 1, float(eps), stack_route if l == 0 else 0)   # pointers: NULL checks only
    return bool(lib.kgcn_gcn_stack_supported(csr.rows, csr.max_nnz, arr, len(spec)))


class _GcnStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, csr, enabled, spec, buffers, gather, *params):
        import ctypes
        x = _f32c(x, "inputs")
        T, N, d0 = x.shape
        if (T, N) != (csr.num_graphs, csr.rows) or d0 != spec[0][2]:
            raise _lib.KgcnHipError("inputs %s do not match the adjacency batch / the first layer" % (tuple(x.shape),))
        arr, keep = _stack_descriptors(spec, params, buffers)
        outs = [torch.empty((T, N, s[3]), device=x.device, dtype=torch.float32) for s in spec]
        optr = (ctypes.c_void_p * len(spec))(*[o.data_ptr() for o in outs])
        pooled = torch.empty((T, spec[-1][3]), device=x.device, dtype=torch.float32) if gather else None
        check(lib.kgcn_gcn_stack_fwd_f32(csr.desc(), ptr(x), ptr(enabled), arr, len(spec), optr, ptr(pooled), current_stream()),
              "kgcn_gcn_stack_fwd_f32")
        ctx.csr, ctx.enabled, ctx.spec, ctx.buffers, ctx.gather = csr, enabled, spec, buffers, gather
        ctx.nparams = len(params)
        ctx.save_for_backward(x, *outs, *[p for p in params if p is not None])
        ctx.param_none = [p is None for p in params]
        ctx.param_shapes = [None if p is None else tuple(p.shape) for p in params]
        # (the gradients are views of ONE buffer nothing reads inside the pass: its second stage may wait -- see deferred_reductions)
        ctx.defer_ok = all(p is None or p.is_leaf for p in params)
        ctx.defer_ids = tuple(params)
        _count_use(*params)
        return pooled if gather else outs[-1]

    @staticmethod
    def backward(ctx, g):
        import ctypes
        spec = ctx.spec
        saved = list(ctx.saved_tensors)
        x, outs, rest = saved[0], saved[1:1 + len(spec)], saved[1 + len(spec):]
        params, it = [], iter(rest)
        for is_none in ctx.param_none:
            params.append(None if is_none else next(it))
        g = _f32c(g, "grad")
        arr, keep = _stack_descriptors(spec, params, ctx.buffers)
        optr = (ctypes.c_void_p * len(spec))(*[o.data_ptr() for o in outs])
        n = int(lib.kgcn_gcn_stack_param_floats(arr, len(spec)))
        dparams = torch.empty((n,), device=g.device, dtype=torch.float32)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        wsb = lib.kgcn_gcn_stack_bwd_workspace_bytes(x.shape[0], arr, len(spec))
        ws = torch.empty((max(wsb, 4) // 4,), device=g.device, dtype=torch.float32)
        with _no_deferral_unless(ctx.defer_ok and _single_use(*ctx.defer_ids)):
            check(lib.kgcn_gcn_stack_bwd_f32(ctx.csr.transpose().desc(), ptr(x), ptr(ctx.enabled), arr, len(spec), optr, ptr(g),
                                             1 if ctx.gather else 0, ptr(dx), ptr(dparams), ptr(ws), wsb, current_stream()),
                  "kgcn_gcn_stack_bwd_f32")
            _keep_until_flush(ws)
        grads, off = [], 0
        for l, (kind, act, din, dout, eps) in enumerate(spec):
            nw = dout if kind == 2 else din * dout
            for which, cnt in ((0, nw), (1, dout)):
                i = 2 * l + which
                if ctx.param_none[i]:
                    grads.append(None)
                else:
                    grads.append(dparams[off:off + cnt].reshape(ctx.param_shapes[i]))
                off += cnt
        return (dx, None, None, None, None, None) + tuple(grads)


def gcn_stack(x, csr, enabled, spec, buffers, gather, params):
    """x [T, N, d0]; spec / buffers / params as in _stack_descriptors.  Returns pooled [T, d_last] (gather) or the last
    layer's output [T, N, d_last]."""
    return _GcnStack.apply(x, csr, enabled, tuple(spec), tuple(buffers), bool(gather), *params)


# -------------------------------------------------------------------------------------------------
# aggregate-first GraphConv operand: [X | 1 | 0..]
# -------------------------------------------------------------------------------------------------
class _AugmentOnes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, width):
        x2d = _f32c(x2d, "inputs")
        m, din = x2d.shape
        out = torch.empty((m, width), device=x2d.device, dtype=torch.float32)
        check(lib.kgcn_augment_ones_f32(ptr(x2d), m, din, din, ptr(out), width, current_stream()), "kgcn_augment_ones_f32")
        ctx.shape = (m, din)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g, "grad")
        m, din = ctx.shape
        dx = torch.empty((m, din), device=g.device, dtype=torch.float32)
        check(lib.kgcn_augment_ones_bwd_f32(ptr(g), m, din, g.shape[1], ptr(dx), din, current_stream()),
              "kgcn_augment_ones_bwd_f32")
        return dx, None


def augment_ones(x2d, width):
    """[m, din] -> [m, width] = [x | 1 | 0 ...] (width >= din + 1)."""
    return _AugmentOnes.apply(x2d, width)


# -------------------------------------------------------------------------------------------------
# losses of the model files: one HIP pass (cost per graph, d cost_sum / d logits, partial sums) + one finishing block
# -------------------------------------------------------------------------------------------------
def _loss_grad(dlog, g_opt, g_sum, batch):
    """d logits = dlog * (g_sum + g_opt / batch) in one launch (kgcn_loss_grad_f32)."""
    if g_opt is None and g_sum is None:
        return None
    go = None if g_opt is None else _f32c(g_opt.reshape(1), "grad")
    gs = None if g_sum is None else _f32c(g_sum.reshape(1), "grad")
    out = torch.empty_like(dlog)
    check(lib.kgcn_loss_grad_f32(ptr(dlog), ptr(go), ptr(gs), batch, dlog.numel(), ptr(out), current_stream()), "kgcn_loss_grad_f32")
    return out


_pos_weight_cache = {}         # id(host object) -> (the object itself, content bytes, device, device tensor)


def _pos_weight_operand(pos_weight, W, device):
    """-> (scalar weight, per-task device tensor or None).  A HOST per-task pos_weight (list / numpy array / CPU tensor -- the
    reference's info.pos_weight is a numpy array, kgcn/data_util.py:563-568) is uploaded ONCE and cached on its content: a
    per-step torch.as_tensor(...).to(device) is a synchronous pageable copy every forward and an illegal operation inside a
    hipGraph capture.  Device tensors are used where they lie (a 1-element one is broadcast on the device: no float() sync)."""
    if pos_weight is None:
        return 0.0, None
    if torch.is_tensor(pos_weight) and pos_weight.is_cuda:
        qv = pos_weight.detach().to(torch.float32).reshape(-1)
        if qv.numel() == 1:
            qv = qv.expand(W)
        if qv.numel() != W:
            raise _lib.KgcnHipError("pos_weight has %d entries for %d tasks" % (qv.numel(), W))
        return 0.0, qv.contiguous()
    if not (torch.is_tensor(pos_weight) or hasattr(pos_weight, "__len__")):
        return float(pos_weight), None
    host = torch.as_tensor(pos_weight, dtype=torch.float32).reshape(-1)
    if host.numel() == 1:
        return float(host), None
    if host.numel() != W:
        raise _lib.KgcnHipError("pos_weight has %d entries for %d tasks" % (host.numel(), W))
    content = host.numpy().tobytes()
    hit = _pos_weight_cache.get(id(pos_weight))
    if hit is not None and hit[0] is pos_weight and hit[1] == content and hit[2] == device:
        return 0.0, hit[3]
    if torch.cuda.is_current_stream_capturing():
        raise _lib.KgcnHipError("a host pos_weight was first seen (or changed) while a hipGraph is being captured: pass a device "
                                "tensor, or run one eager step with this pos_weight before the capture")
    if len(_pos_weight_cache) >= 64:
        _pos_weight_cache.clear()
    qv = host.to(device).contiguous()
    _pos_weight_cache[id(pos_weight)] = (pos_weight, content, device, qv)
    return 0.0, qv


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, mask, mask_label, kind, pos_weight):
        x = _f32c(logits, "logits")
        if x.dim() != 2:
            raise _lib.KgcnHipError("logits must be [batch, width], got %s" % (tuple(x.shape),))
        B, W = x.shape
        z = _f32c(labels.to(torch.float32), "labels")
        mk = _f32c(mask.to(torch.float32).reshape(-1), "mask")
        ml = None if mask_label is None else _f32c(mask_label.to(torch.float32), "mask_label")
        if tuple(z.shape) != (B, W) or mk.numel() != B or (ml is not None and tuple(ml.shape) != (B, W)):
            raise _lib.KgcnHipError("labels / mask / mask_label do not match logits %s" % (tuple(x.shape),))
        dlog = torch.empty_like(x)
        sums = torch.empty((2,), device=x.device, dtype=torch.float32)
        wsb = lib.kgcn_loss_workspace_bytes(B)
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        if kind == "sigmoid":
            # info.pos_weight of the reference is one weight per label column (kgcn/data_util.py:563-568, consumed by
            # example_model/model_multitask.py:72-76); a scalar is accepted too and applies to every task
            qs, qv = _pos_weight_operand(pos_weight, W, x.device)
            check(lib.kgcn_masked_sigmoid_ce_f32(ptr(x), ptr(z), ptr(mk), ptr(ml), B, W, 0 if pos_weight is None else 1,
                                                 qs, ptr(qv), None, ptr(dlog), ptr(sums), ptr(ws), wsb, current_stream()),
                  "kgcn_masked_sigmoid_ce_f32")
        else:
            check(lib.kgcn_masked_softmax_ce_f32(ptr(x), ptr(z), ptr(mk), B, W, None, ptr(dlog), ptr(sums), ptr(ws), wsb,
                                                 current_stream()), "kgcn_masked_softmax_ce_f32")
        ctx.save_for_backward(dlog)
        ctx.batch = B
        ctx.set_materialize_grads(False)           # an output nobody differentiates must not become a zero-filled gradient
        return sums[1], sums[0]                    # cost_opt (mean over the padded batch, Q5), cost_sum

    @staticmethod
    def backward(ctx, g_opt, g_sum):
        (dlog,) = ctx.saved_tensors
        return _loss_grad(dlog, g_opt, g_sum, ctx.batch), None, None, None, None, None


class _SparseCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label_idx):
        x = _f32c(logits, "logits")
        B, W = x.shape
        idx = label_idx.reshape(-1).to(torch.int64).contiguous()
        require_gpu(idx, "labels")
        if idx.numel() != B:
            raise _lib.KgcnHipError("labels have %d entries for %d rows of logits" % (idx.numel(), B))
        dlog = torch.empty_like(x)
        sums = torch.empty((2,), device=x.device, dtype=torch.float32)
        wsb = lib.kgcn_loss_workspace_bytes(B)
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device, dtype=torch.float32)
        check(lib.kgcn_sparse_softmax_ce_f32(ptr(x), ptr(idx), None, B, W, None, ptr(dlog), ptr(sums), ptr(ws), wsb,
                                             current_stream()), "kgcn_sparse_softmax_ce_f32")
        ctx.save_for_backward(dlog)
        return sums[0]

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return _loss_grad(dlog, None, g, 1), None          # dlog * g in the library's launch (no ATen kernel in the step)


def sparse_softmax_ce_sum(logits, label_idx):
    """example_model/sparse.py:112-113: reduce_sum(sparse_softmax_cross_entropy_with_logits(labels, logits))."""
    return _SparseCE.apply(logits, label_idx)


def masked_sigmoid_ce(logits, labels, mask, mask_label=None, pos_weight=None):
    """example_model/model_multitask.py:66-79 -> (cost_opt, cost_sum)."""
    return _MaskedCE.apply(logits, labels, mask, mask_label, "sigmoid", pos_weight)


def masked_softmax_ce(logits, labels, mask):
    """example_model/model.py:56-61 -> (cost_opt, cost_sum)."""
    return _MaskedCE.apply(logits, labels, mask, None, "softmax", None)


__all__ = ["BatchedCSR", "BatchedAdjacency", "bspmm", "bspmm_raw", "bconv", "dense", "activation", "act_code",
           "graphconv_fused", "graphconv_fused_supported", "gin_aggregate", "graph_gather",
           "graph_maxpool", "gat", "gram", "ragged_gather", "ragged_compact_rows",
           "masked_sigmoid_ce", "masked_softmax_ce", "sparse_softmax_ce_sum", "augment_ones",
           "gcn_stack", "gcn_stack_supported", "graph_gather_tee"]


# -------------------------------------------------------------------------------------------------
# GraphBatchNormalization (kgcn/layers.py:170-220): statistics / apply / backward over the valid node rows
# -------------------------------------------------------------------------------------------------
def graph_bn_stats(x, enabled=None):
    """Batch mean and (population) variance per feature over the valid rows of x [T, N, D] (training mode)."""
    x = _f32c(x, "inputs")
    T, N, D = x.shape
    mean = torch.empty((D,), device=x.device, dtype=torch.float32)
    var = torch.empty_like(mean)
    wsb = lib.kgcn_graph_bn_workspace_bytes(D)
    wsp = torch.empty((wsb // 4 + 1,), device=x.device, dtype=torch.float32)
    check(lib.kgcn_graph_bn_stats_f32(ptr(x), T, N, D, ptr(enabled), ptr(mean), ptr(var), ptr(wsp), wsb,
                                      current_stream()), "kgcn_graph_bn_stats_f32")
    return mean, var


class _GraphBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, enabled, eps, training, act=0):
        x = _f32c(x, "inputs")
        T, N, D = x.shape
        # learning phase 0: d gamma / d beta are not read inside the backward call (the C side queues their second stage in a deferral scope)
        ctx.defer_ok = bool(not training and gamma.is_leaf and beta.is_leaf)
        ctx.defer_ids = (gamma, beta)
        _count_use(gamma, beta)
        gamma, beta = _f32c(gamma, "gamma"), _f32c(beta, "beta")
        y = torch.empty_like(x)
        check(lib.kgcn_graph_bn_apply_act_f32(ptr(x), T, N, D, ptr(enabled), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                              float(eps), int(act), ptr(y), current_stream()), "kgcn_graph_bn_apply_act_f32")
        ctx.save_for_backward(x, gamma, mean, var, y if act else x)
        ctx.enabled, ctx.eps, ctx.training, ctx.act = enabled, float(eps), bool(training), int(act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, var, yact = ctx.saved_tensors
        g = _f32c(g, "grad")
        T, N, D = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgamma = torch.empty((D,), device=x.device, dtype=torch.float32)
        dbeta = torch.empty_like(dgamma)
        wsb = lib.kgcn_graph_bn_workspace_bytes(D)
        wsp = torch.empty((wsb // 4 + 1,), device=x.device, dtype=torch.float32)
        # d gamma / d beta receive their second stage at the flush: both must still be alive then.  autograd drops the
        # gradient of an input that does not require one as soon as backward returns (frozen affine parameters, a
        # detached gamma of the inference call), so deferral needs BOTH to be wanted: AccumulateGrad then takes the two
        # tensors themselves as .grad (deferred_reductions excludes parameters whose .grad exists or that carry hooks).
        # They must NOT be referenced from here as well: a second reference makes AccumulateGrad clone them -- before the
        # flush has written them (and costs two copy launches per step).
        want_affine = bool(ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
        with _no_deferral_unless(ctx.defer_ok and want_affine and dx is not None and _single_use(*ctx.defer_ids)):
            check(lib.kgcn_graph_bn_bwd_dact_f32(ptr(x), ptr(g), ptr(yact) if ctx.act else None, ctx.act, T, N, D,
                                                 ptr(ctx.enabled), ptr(mean), ptr(var), ptr(gamma), ctx.eps, int(ctx.training),
                                                 ptr(dx), ptr(dgamma), ptr(dbeta), ptr(wsp), wsb, current_stream()),
                  "kgcn_graph_bn_bwd_dact_f32")
            _keep_until_flush(wsp)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def graph_bn(x, gamma, beta, mean, var, enabled=None, eps=1e-3, training=False, activation=None):
    """y = act(gamma (x - mean) / sqrt(var + eps) + beta on the valid rows, 0 on the padding rows).  training=True: mean /
    var are THIS batch's statistics and the backward differentiates through them.  The activation rides in the same pass
    (and its derivative in the backward's reads)."""
    return _GraphBN.apply(x, gamma, beta, mean, var, enabled, eps, training, act_code(activation))
