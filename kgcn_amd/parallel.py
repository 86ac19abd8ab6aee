"""Data-parallel glue for the hot path: graphs are independent units, so a batch shards across
ranks with NO communication in forward/backward; the only exchange is one all-reduce of the
(tiny: 16-260 KB) parameter-gradient bucket per step -- RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).  The reference has no
multi-device path at all; this is the build's addition.

The payload is latency-bound (tens of microseconds), so everything goes in ONE flat fp32 bucket
and one collective.  The reference's loss is reduce_mean over the PADDED batch (quirk Q5,
example_model/model.py:58-61): rank r's local mean over its B_r padded graphs enters the global mean
with weight B_r / sum(B) -- `shard_weight` -- so unequal shards (shard_range hands out sizes that
differ by one) still reproduce the single-process gradients; equal shards reduce to the plain mean.
"""
import torch
import torch.distributed as dist


def shard_range(num_graphs, rank, world_size):
    """Contiguous shard [lo, hi) of a global batch for `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(num_graphs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_weight(local_graphs, global_graphs):
    """Weight of this rank's LOCAL-mean gradients in the global mean over the padded batch (Q5)."""
    if global_graphs <= 0:
        raise ValueError("global batch is empty")
    return float(local_graphs) / float(global_graphs)


class GradBucket:
    """One flat fp32 gradient bucket for a fixed parameter list."""

    def __init__(self, params, flat=None):
        """flat: a kgcn_amd.train.FlatParameters over the same parameters (TFAdam(...).flat) -- the bucket then IS its flat
        gradient buffer: the all-reduce runs on the buffer the fused optimiser update reads, nothing is copied back."""
        self.flat_params = flat
        if flat is not None:
            params = flat.params
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("GradBucket: empty parameter list -- the layers create their parameters on the first "
                             "forward pass (Keras build semantics, kgcn/layers.py:48-62); run one forward or call "
                             "layer.build(input_shape) before collecting model.parameters()")
        self.sizes = [p.numel() for p in self.params]
        self.total = sum(self.sizes) if flat is None else flat.total
        self._flat = None

    def _buffer(self, like):
        if self.flat_params is not None:
            return self.flat_params.grad
        if self._flat is None or self._flat.device != like.device:
            self._flat = torch.empty((self.total,), dtype=torch.float32, device=like.device)
        return self._flat

    @staticmethod
    def _contiguous_run(grads):
        """The gradients as ONE flat tensor if they are contiguous, consecutive slices of the same storage (in order); else None."""
        g0 = grads[0]
        if g0.dtype != torch.float32 or not all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
            return None
        try:
            base = g0.untyped_storage().data_ptr()
        except Exception:                                        # noqa: BLE001
            return None
        nxt = g0.data_ptr()
        for g in grads:
            if g.untyped_storage().data_ptr() != base or g.data_ptr() != nxt:
                return None
            nxt += g.numel() * 4
        total = sum(g.numel() for g in grads)
        return torch.as_strided(g0, (total,), (1,), g0.storage_offset()) if len(grads) > 1 else g0.view(-1)

    def _views(self, flat):
        if self.flat_params is not None:
            return self.flat_params.grad_views
        out, off = [], 0
        for p, n in zip(self.params, self.sizes):
            out.append(flat[off:off + n].view(p.shape))
            off += n
        return out

    @staticmethod
    def _use_avg(world, weight, group):
        """equal shards on RCCL: the mean is the collective's own reduction (AVG), no scale launch; gloo has no AVG.  Depends only
        on (world, weight, backend) -- identical on every rank."""
        return bool(world > 1 and weight is None and dist.is_initialized() and dist.get_backend(group) == "nccl")

    def all_reduce_mean(self, group=None, weight=None, unpack=True):
        """flat <- concat(grads); all_reduce; scatter back into .grad (in place).  weight=None: plain mean over the
        ranks (equal shards).  weight=w_r (shard_weight: local padded graphs / global padded graphs): sum_r w_r g_r,
        the gradient of the reference's reduce_mean over the global padded batch for shards of any size.
        unpack=False: the reduced gradients stay in the flat buffer (the fused optimiser update reads it there).
        Four launches per step whatever the number of parameters: one multi-tensor pack, one scale, the collective,
        one multi-tensor unpack -- the payload is latency-bound, so launches are what it costs.  Capturable: inside
        a hipGraph capture the collective is recorded on the capturing stream like any kernel."""
        if torch.cuda.is_available():
            from . import ops
            ops.join_side_streams()
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("GradBucket: a parameter has no gradient")
        run = self._contiguous_run(grads) if self.flat_params is None else None
        if run is not None:
            # the backward already left the gradients as consecutive slices of one buffer (the fused GraphConv backward writes
            # [dW | dbias] into one allocation): scale and all-reduce THAT -- the pack and unpack launches disappear
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            if world == 1 and weight is not None and float(weight) != 1.0:
                raise ValueError("a single rank owns the whole batch: its weight must be 1")
            avg = self._use_avg(world, weight, group)
            if world > 1 and not avg:
                run.mul_(1.0 / world if weight is None else float(weight))
            if dist.is_initialized():
                dist.all_reduce(run, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group)
            return run
        flat = self._buffer(grads[0])
        views = self._views(flat)
        torch._foreach_copy_(views, grads)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if world == 1 and weight is not None and float(weight) != 1.0:
            raise ValueError("a single rank owns the whole batch: its weight must be 1")
        # ONE formulation whichever buffer is reduced: which of the two paths a rank takes depends on the layout its own backward
        # left the gradients in (kernel routes can differ between ranks: graphconv_fused_supported looks at the LOCAL batch) --
        # the collective's reduction op and the pre-scale may only depend on what every rank agrees on (world, weight, backend)
        avg = self._use_avg(world, weight, group)
        if world > 1 and not avg:
            flat.mul_(1.0 / world if weight is None else float(weight))
        if dist.is_initialized():
            # also with ONE rank: the collective is a no-op numerically but goes through RCCL (and, inside a hipGraph
            # capture, into the graph) -- the only way a 1-GPU box exercises the data-parallel path end to end
            dist.all_reduce(flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group)
        if unpack:
            torch._foreach_copy_(grads, views)
        return flat
