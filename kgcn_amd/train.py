"""Minimal train step for the model restatements: TF-style Adam and one
mini-batch step as CoreModel.fit runs it (kgcn/core.py:121-127, 257-270).  Not the reference's
session harness (checkpoints, early stopping, metrics files are out of scope)."""
import torch


class TFAdam:
    """tf.train.AdamOptimizer(lr) (kgcn/core.py:124), defaults beta1 .9, beta2 .999, eps 1e-8:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  p -= lr_t * m / (sqrt(v) + eps)
    -- "epsilon hat" OUTSIDE the bias-corrected square root, unlike torch.optim.Adam."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, capturable=False):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("TFAdam: empty parameter list -- the layers create their parameters on the first forward "
                             "pass (Keras build semantics); run one forward before collecting model.parameters()")
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        # capturable: the step counter lives on the device, so that a captured hipGraph advances it
        self.capturable = capturable
        self._t_dev = torch.zeros((), dtype=torch.float64, device=self.params[0].device) if capturable else None

    def zero_grad(self, set_to_none=True):
        if set_to_none or any(p.grad is None for p in self.params):
            for p in self.params:
                p.grad = None
        else:
            torch._foreach_zero_([p.grad for p in self.params])

    @torch.no_grad()
    def step(self):
        """One update of every parameter with multi-tensor (_foreach) ops: 8 launches for the whole model
        instead of 9 per parameter -- at the reference's batch size the step is launch-bound."""
        self.t += 1
        grads = [p.grad for p in self.params]
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, grads, alpha=1.0 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, grads, grads, value=1.0 - self.b2)
        denom = torch._foreach_sqrt(self.v)
        torch._foreach_add_(denom, self.eps)
        upd = torch._foreach_div(self.m, denom)
        if self.capturable:
            self._t_dev += 1
            lr_t = (self.lr * torch.sqrt(1.0 - self.b2 ** self._t_dev) / (1.0 - self.b1 ** self._t_dev)).float()
            torch._foreach_mul_(upd, lr_t)
            torch._foreach_sub_(self.params, upd)
        else:
            lr_t = self.lr * (1.0 - self.b2 ** self.t) ** 0.5 / (1.0 - self.b1 ** self.t)
            torch._foreach_add_(self.params, upd, alpha=-lr_t)


def train_step(model, optimizer, loss_fn, features, adjs, labels, mask, bucket=None, shard_weight=None,
               **fwd_kwargs):
    """sess.run([train_step, cost_sum]) of one mini-batch; `bucket` (kgcn_amd.parallel.GradBucket)
    combines the gradients of the data-parallel ranks before the update (shard_weight: this rank's share
    of the global padded batch, kgcn_amd.parallel.shard_weight; None = equal shards)."""
    optimizer.zero_grad()
    logits = model(features, adjs, **fwd_kwargs)
    cost_opt, cost_sum = loss_fn(logits, labels, mask)
    cost_opt.backward()
    if bucket is not None:
        bucket.all_reduce_mean(weight=shard_weight)
    optimizer.step()
    return float(cost_sum.detach()), logits.detach()


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
                 shard_weight=None, **fwd_kwargs):
        """bucket / shard_weight: data parallel -- the ONE all-reduce of the flat gradient bucket (RCCL) is captured
        in the graph between backward and the Adam update, so a replay is still a single host call per step."""
        if not optimizer.capturable:
            raise ValueError("GraphedTrainStep needs TFAdam(capturable=True)")
        self.bucket, self.shard_weight = bucket, shard_weight
        self.model, self.opt, self.loss_fn, self.sb = model, optimizer, loss_fn, static_batch
        self.labels, self.mask, self.kw = labels, mask, fwd_kwargs
        self.cost_sum = self.logits = None
        # warm-up and capture must not train: model and optimiser state are restored afterwards
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
        with torch.cuda.graph(self.graph):
            self._eager()
        with torch.no_grad():
            for dst, src in zip(list(optimizer.params) + optimizer.m + optimizer.v, saved):
                dst.copy_(src)
            optimizer._t_dev.fill_(saved_t)
        optimizer.t = saved_t
        if hasattr(static_batch, "prune_unused"):
            static_batch.prune_unused()

    def _eager(self):
        self.opt.zero_grad(set_to_none=False)
        logits = self.model(self.sb.features, self.sb.adjacency, **self.kw)
        cost_opt, cost_sum = self.loss_fn(logits, self.labels, self.mask)
        cost_opt.backward()
        if self.bucket is not None:
            self.bucket.all_reduce_mean(weight=self.shard_weight)
        self.opt.step()
        self.cost_sum, self.logits = cost_sum.detach(), logits.detach()

    def replay(self):
        """Runs the captured step on whatever the static buffers hold now; returns (cost_sum, logits)
        device tensors (no synchronisation)."""
        self.graph.replay()
        self.opt.t += 1
        return self.cost_sum, self.logits
