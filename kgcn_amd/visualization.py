"""Attribution loop of the reference's visualisation, on the HIP path.

kgcn/visualization.py:187-260 (CompoundVisualizer.cal_integrated_gradients): the inputs named in
`perturbation_target` (node features and / or the values of adjacency channel 0) are scaled by k/D for
k = 1..D, the gradient of the target prediction with respect to the `ig_modal_target` inputs is taken at
every step, and IG[modal] += grad * data / D; "grad_prod" and "grad" are the one-step variants.  The
consumer of the d values gradient of the batched SpMM (kgcn/bspmm_call.py:50-55) is exactly this loop.

score_fn(features, adjacency) -> scalar tensor (e.g. one softmax probability of one graph); `adjacency`
is a kgcn_amd.BatchedAdjacency whose channel-0 values arrive as a differentiable tensor
(BatchedAdjacency.with_values), features a [B, N, F] tensor.
"""
import torch


def integrated_gradients(score_fn, features, adjacency, divide_number=100, modal=("features", "adjs"),
                         perturbation=None, method="ig"):
    """Returns {"features": [B, N, F] tensor, "adjs": [nnz] tensor in the CSR order of channel 0,
    "sum_of_ig": float, "start_score": f(scale 0), "end_score": f(scale 1)} (only the requested modals).
    For method "ig" the completeness check of the reference (:262-275) is sum_of_ig ~ end - start."""
    modal = tuple(modal)
    pert = modal if perturbation is None else tuple(perturbation)
    base_vals = [c.values for c in adjacency.channels]
    x0 = features.detach()
    ig = {}
    if "features" in modal:
        ig["features"] = torch.zeros_like(x0)
    if "adjs" in modal:
        ig["adjs"] = torch.zeros_like(base_vals[0])

    def grads_at(scale):
        x = (x0 * scale if "features" in pert else x0).clone().requires_grad_("features" in modal)
        vals = [v.clone() for v in base_vals]
        if "adjs" in pert:
            vals[0] = vals[0] * scale
        vals[0] = vals[0].requires_grad_("adjs" in modal)
        score = score_fn(x, adjacency.with_values(vals))
        wrt = ([x] if "features" in modal else []) + ([vals[0]] if "adjs" in modal else [])
        g = torch.autograd.grad(score, wrt)
        out = {}
        if "features" in modal:
            out["features"] = g[0]
        if "adjs" in modal:
            out["adjs"] = g[-1]
        return out, float(score.detach())

    data = {"features": x0, "adjs": base_vals[0]}
    if method == "ig":
        for k in range(divide_number):
            g, _ = grads_at((k + 1) / float(divide_number))
            for m in ig:
                ig[m] += g[m] * data[m] / float(divide_number)
    elif method in ("grad_prod", "grad"):
        g, _ = grads_at(1.0)
        for m in ig:
            ig[m] += g[m] * data[m] if method == "grad_prod" else g[m]
    else:
        raise ValueError("unsupported method %r (ig, grad_prod, grad)" % (method,))
    with torch.no_grad():
        xs = x0 * 0.0 if "features" in pert else x0
        vs = [v.clone() for v in base_vals]
        if "adjs" in pert:
            vs[0] = vs[0] * 0.0
        start = float(score_fn(xs, adjacency.with_values(vs)))
        end = float(score_fn(x0, adjacency.with_values(base_vals)))
    res = dict(ig)
    res["sum_of_ig"] = float(sum(v.sum() for v in ig.values()))
    res["start_score"], res["end_score"] = start, end
    return res


def values_to_dense(csr, values):
    """sparse_to_dense_core of the reference (:208): per-entry values of channel 0 -> dense [T, M, K]."""
    rp = csr.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(rp.numel() - 1, device=rp.device), rp[1:] - rp[:-1])
    dense = torch.zeros((csr.num_graphs * csr.rows, csr.cols), device=values.device, dtype=values.dtype)
    dense.index_put_((rows, csr.cv[:, 0].long()), values, accumulate=True)
    return dense.reshape(csr.num_graphs, csr.rows, csr.cols)
