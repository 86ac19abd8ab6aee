"""Minimal train step for the model restatements: TF-style Adam and one
mini-batch step as CoreModel.fit runs it (kgcn/core.py:121-127, 257-270).  Not the reference's
session harness (checkpoints, early stopping, metrics files are out of scope)."""
import torch


class FlatParameters:
    """All parameters of a model in ONE flat fp32 buffer (each tensor at a 256-byte aligned offset, the nn.Parameters
    re-pointed at views of it) next to a same-shaped flat gradient buffer: the optimiser update is one kernel over the
    buffer, and the data-parallel exchange all-reduces the gradient buffer as it is (kgcn_amd.parallel.GradBucket shares
    it) -- no per-tensor launches.  The parameters keep their identity (model.parameters() still yields them); moving the
    model to another device afterwards would break the sharing."""
    ALIGN = 64                                   # floats

    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatParameters: empty parameter list -- the layers create their parameters on the first forward "
                             "pass (Keras build semantics); run one forward before collecting model.parameters()")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise ValueError("the fused optimiser needs device-resident parameters (kgcn_amd has no CPU path)")
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be float32 on one device")
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.total = off
        self.data = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.views, self.grad_views = [], []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.data[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.views.append(v)
                self.grad_views.append(self.grad[o:o + p.numel()].view(p.shape))

    def pack_grads(self):
        """flat gradient buffer <- the parameters' .grad tensors (ONE multi-tensor copy launch)."""
        from . import ops
        ops.join_side_streams()                 # weight gradients of the big layers may still run on the side stream
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("FlatParameters: a parameter has no gradient")
        torch._foreach_copy_(self.grad_views, grads)
        return self.grad


class TFAdam:
    """tf.train.AdamOptimizer(lr) (kgcn/core.py:124), defaults beta1 .9, beta2 .999, eps 1e-8:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  p -= lr_t * m / (sqrt(v) + eps)
    -- "epsilon hat" OUTSIDE the bias-corrected square root, unlike torch.optim.Adam.
    The update of the WHOLE model is one HIP kernel over the flat parameter buffer (kgcn_adam_tf_f32, csrc/train.hip); the
    step counter lives in device memory, so a captured hipGraph advances it on replay (`capturable` is accepted for
    compatibility: the update is always capturable)."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, capturable=True):
        params = [p for p in params]
        if not params:
            raise ValueError("TFAdam: empty parameter list -- the layers create their parameters on the first forward "
                             "pass (Keras build semantics); run one forward before collecting model.parameters()")
        self.flat = FlatParameters(params)
        self.params = self.flat.params
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self._m = torch.zeros_like(self.flat.data)
        self._v = torch.zeros_like(self.flat.data)
        # per-parameter views of the moment buffers (state inspection / save-restore)
        self.m = [self._m[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.flat.offsets)]
        self.v = [self._v[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.flat.offsets)]
        self.capturable = True
        self._t_dev = torch.zeros((), dtype=torch.int64, device=self.flat.data.device)

    def zero_grad(self, set_to_none=True):
        if set_to_none or any(p.grad is None for p in self.params):
            for p in self.params:
                p.grad = None
        else:
            torch._foreach_zero_([p.grad for p in self.params])

    @torch.no_grad()
    def step(self, packed=False):
        """One update of every parameter: the update kernel + the counter tick.  packed: the flat gradient buffer already holds the
        (exchanged) gradients -- a GradBucket sharing it did the pack; otherwise the kernel reads each parameter's .grad tensor in
        place (kgcn_adam_tf_multi_f32), no packed copy."""
        from . import _lib
        from . import ops
        ops.weight_tables.invalidate()             # the update rewrites the weights behind torch's version counters
        self.t += 1
        if not packed:
            # no exchange in front of the update: the kernel reads every gradient where the backward pass left it -- no packed copy
            import ctypes
            ops.join_side_streams()
            segs = (_lib.AdamSegment * len(self.params))()
            keep = []
            for i, (p, o) in enumerate(zip(self.params, self.flat.offsets)):
                g = p.grad
                if g is None:
                    raise RuntimeError("TFAdam: a parameter has no gradient")
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.to(torch.float32).contiguous()
                    keep.append(g)
                segs[i] = _lib.AdamSegment(g.data_ptr(), o, p.numel())
            _lib.check(_lib.lib.kgcn_adam_tf_multi_f32(_lib.ptr(self.flat.data), _lib.ptr(self._m), _lib.ptr(self._v),
                                                       self.flat.total, ctypes.cast(segs, ctypes.c_void_p), len(self.params),
                                                       float(self.lr), float(self.b1), float(self.b2), float(self.eps),
                                                       _lib.ptr(self._t_dev), _lib.current_stream()), "kgcn_adam_tf_multi_f32")
            return
        _lib.check(_lib.lib.kgcn_adam_tf_f32(_lib.ptr(self.flat.data), _lib.ptr(self.flat.grad), _lib.ptr(self._m),
                                             _lib.ptr(self._v), self.flat.total, float(self.lr), float(self.b1),
                                             float(self.b2), float(self.eps), _lib.ptr(self._t_dev), _lib.current_stream()),
                   "kgcn_adam_tf_f32")


def _exchange(bucket, optimizer, shard_weight):
    """Data-parallel gradient exchange in front of the update; returns True when the flat gradient buffer of the optimiser
    already holds the reduced gradients (bucket built on optimizer.flat)."""
    if bucket is None:
        return False
    shared = getattr(bucket, "flat_params", None) is optimizer.flat
    bucket.all_reduce_mean(weight=shard_weight, unpack=not shared)
    return shared


def train_step(model, optimizer, loss_fn, features, adjs, labels, mask, bucket=None, shard_weight=None,
               **fwd_kwargs):
    """sess.run([train_step, cost_sum]) of one mini-batch; `bucket` (kgcn_amd.parallel.GradBucket)
    combines the gradients of the data-parallel ranks before the update (shard_weight: this rank's share
    of the global padded batch, kgcn_amd.parallel.shard_weight; None = equal shards)."""
    from . import ops
    optimizer.zero_grad()
    ops.weight_tables.refresh()                    # every wide layer's W / W^T fragment tables in one launch
    logits = model(features, adjs, **fwd_kwargs)
    cost_opt, cost_sum = loss_fn(logits, labels, mask)
    # .grad is None here (zero_grad above): backward's result tensors BECOME the gradients, nothing reads them before the block
    # ends -- so the second stages of all weight gradients can wait for one launch (ops.deferred_reductions)
    with ops.deferred_reductions(root=cost_opt):
        cost_opt.backward()
    optimizer.step(packed=_exchange(bucket, optimizer, shard_weight))
    return float(cost_sum.detach()), logits.detach()


def capture_mode():
    """Keyword arguments for torch.cuda.graph().  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events of
    the collectives enqueued before the capture (the warm-up steps' all-reduces, the barrier) every ~100 ms; under the default
    capture_error_mode="global" such a hipEventQuery from another thread invalidates the capture ("operation not permitted when stream
    is capturing": seen once in fifteen one-rank RCCL runs of cfg4, profiles/r06_capture_vs_watchdog.txt).  The device is idle at this
    point (the caller has synchronised), so one watchdog period later its work list is empty; and the capture only polices its OWN
    thread (the all-reduce of the step is issued by the capturing thread and is captured like any kernel)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        import time
        time.sleep(0.25)
        return {"capture_error_mode": "thread_local"}
    return {}


class GraphedTrainStep:
    """One mini-batch step (forward, loss, backward, TF-Adam update) captured ONCE in a hipGraph and
    replayed per batch.  At the reference's own batch size (30 graphs of 10 nodes, example_config/synth.json)
    a step is ~60 tiny kernel launches: launch-bound in eager mode exactly like the reference's B*C tiny TF
    ops -- the graph replays them back to back without host involvement.  Inputs live in
    fixed buffers: a kgcn_amd.data_util.StaticBatch (adjacency + features) and labels / mask tensors that
    the caller overwrites in place before replay().  No tensor of an earlier eager step that still holds
    its autograd graph (a loss, logits) may be alive at construction: the parameters' AccumulateGrad nodes
    would stay bound to the eager stream and break the capture.

    model(features, adjacency, **fwd_kwargs) -> logits;  loss_fn(logits, labels, mask) -> (cost_opt, cost_sum)."""

    def __init__(self, model, optimizer, loss_fn, static_batch, labels, mask, warmup=3, bucket=None,
                 shard_weight=None, capture_assembly=False, **fwd_kwargs):
        """bucket / shard_weight: data parallel -- the ONE all-reduce of the flat gradient bucket (RCCL) is captured
        in the graph between backward and the Adam update, so a replay is still a single host call per step.
        capture_assembly: static_batch.assemble() (mini-batch assembly on the device: adjacency containers, feature rows and
        the tables registered with add_table -- labels, masks) becomes the head of the graph; a step is then
        static_batch.stage(indices) -- one small asynchronous upload -- followed by replay()."""
        if not optimizer.capturable:
            raise ValueError("GraphedTrainStep needs TFAdam(capturable=True)")
        self.bucket, self.shard_weight = bucket, shard_weight
        self.model, self.opt, self.loss_fn, self.sb = model, optimizer, loss_fn, static_batch
        self.labels, self.mask, self.kw = labels, mask, fwd_kwargs
        self.cost_sum = self.logits = self._seed = None
        self.capture_assembly = bool(capture_assembly)
        if self.capture_assembly and not hasattr(static_batch, "assemble"):
            raise ValueError("capture_assembly needs a static batch with stage() / assemble()")
        # warm-up and capture must not train: model and optimiser state are restored afterwards
        if hasattr(static_batch, "reset_usage"):
            static_batch.reset_usage()
        saved = [t.detach().clone() for t in list(optimizer.params) + optimizer.m + optimizer.v]
        saved_t = optimizer.t
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # warm-up on a side stream (allocator, lazy inits)
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **capture_mode()):
            self._eager()
        with torch.no_grad():
            for dst, src in zip(list(optimizer.params) + optimizer.m + optimizer.v, saved):
                dst.copy_(src)
            optimizer._t_dev.fill_(saved_t)
        optimizer.t = saved_t
        if hasattr(static_batch, "prune_unused"):
            static_batch.prune_unused()

    def _eager(self):
        # gradients are NOT accumulated into persistent .grad tensors (one add launch per parameter): .grad is dropped, the
        # backward's own result tensors become the new .grad (static addresses inside the captured graph's memory pool) and
        # one multi-tensor copy packs them into the flat buffer the update kernel / the all-reduce read
        from . import ops
        self.opt.zero_grad(set_to_none=True)
        ops.weight_tables.refresh()                # every wide layer's W / W^T fragment tables in one launch
        if self.capture_assembly:
            self.sb.assemble()
        logits = self.model(self.sb.features, self.sb.adjacency, **self.kw)
        cost_opt, cost_sum = self.loss_fn(logits, self.labels, self.mask)
        if self._seed is None or self._seed.shape != cost_opt.shape:
            self._seed = torch.ones_like(cost_opt)     # persistent d cost / d cost = 1: no fill launch per step
        with ops.deferred_reductions(root=cost_opt):   # one second-stage launch for all weight gradients of the step
            cost_opt.backward(self._seed)
        self.opt.step(packed=_exchange(self.bucket, self.opt, self.shard_weight))
        self.cost_sum, self.logits = cost_sum.detach(), logits.detach()

    def replay(self):
        """Runs the captured step on whatever the static buffers hold now; returns (cost_sum, logits)
        device tensors (no synchronisation)."""
        self.graph.replay()
        self.opt.t += 1
        return self.cost_sum, self.logits
