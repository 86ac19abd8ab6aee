"""Self-check of the model-level oracle (oracle/kgcn_model_oracle.py): hand-written backward vs
finite differences, and that a few Adam steps on synthetic.jbl reduce the loss."""
import numpy as np

from conftest import load_golden, unflatten_adjs
from oracle import kgcn_model_oracle as M


def _batch():
    z = load_golden("g3_synthetic_feed_full30.npz")
    adjs = unflatten_adjs(z, "adj_")
    return z["features"].astype(np.float64), adjs, z["labels"].astype(np.float64), z["mask"].astype(np.float64)


def test_model_backward_finite_difference():
    x, adjs, labels, mask = _batch()
    rng = np.random.default_rng(0)
    p = M.init_params(rng, 3)
    for k in ("b1", "b2", "b3"):
        p[k] = [rng.standard_normal((1, 50)) * 0.1]
    c = M.forward(p, x, adjs, labels, mask)
    g = M.backward(p, c, x, adjs, labels, mask)
    h = 1e-6
    for key, idx in [("w1", (1, 7)), ("w2", (3, 4)), ("w3", (10, 2)), ("b2", (0, 5)), ("dk", (4, 4)),
                     ("db", (3,)), ("ok", (2, 1)), ("ob", (0,)), ("gamma", (6,)), ("beta", (9,))]:
        def at(d):
            q = {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in p.items()}
            t = q[key][0] if isinstance(q[key], list) else q[key]
            t[idx] += d
            return M.forward(q, x, adjs, labels, mask)["cost_opt"]
        fd = (at(h) - at(-h)) / (2 * h)
        an = (g[key][0] if isinstance(g[key], list) else g[key])
        an = np.asarray(an).reshape((p[key][0] if isinstance(p[key], list) else p[key]).shape)[idx]
        assert abs(fd - an) < 1e-7 + 1e-4 * abs(an), (key, fd, an)


def test_adam_steps_reduce_loss():
    x, adjs, labels, mask = _batch()
    p = M.init_params(np.random.default_rng(1), 3)
    opt = M.TFAdam(lr=0.01)
    losses = []
    for _ in range(30):
        p, c = M.train_step(p, opt, x, adjs, labels, mask)
        losses.append(c["cost_opt"])
    assert losses[-1] < losses[0] - 0.01, losses[::5]
