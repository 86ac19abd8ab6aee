"""Minimal train step for the model restatements: TF-style Adam and one
mini-batch step as CoreModel.fit runs it (kgcn/core.py:121-127, 257-270).  Not the reference's
session harness (checkpoints, early stopping, metrics files are out of scope)."""
import torch


class TFAdam:
    """tf.train.AdamOptimizer(lr) (kgcn/core.py:124), defaults beta1 .9, beta2 .999, eps 1e-8:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  p -= lr_t * m / (sqrt(v) + eps)
    -- "epsilon hat" OUTSIDE the bias-corrected square root, unlike torch.optim.Adam."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * (1.0 - self.b2 ** self.t) ** 0.5 / (1.0 - self.b1 ** self.t)
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-lr_t)


def train_step(model, optimizer, loss_fn, features, adjs, labels, mask, bucket=None, **fwd_kwargs):
    """sess.run([train_step, cost_sum]) of one mini-batch; `bucket` (kgcn_amd.parallel.GradBucket)
    averages the gradients over data-parallel ranks before the update."""
    optimizer.zero_grad()
    logits = model(features, adjs, **fwd_kwargs)
    cost_opt, cost_sum = loss_fn(logits, labels, mask)
    cost_opt.backward()
    if bucket is not None:
        bucket.all_reduce_mean()
    optimizer.step()
    return float(cost_sum.detach()), logits.detach()
