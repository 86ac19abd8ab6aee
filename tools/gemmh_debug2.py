"""dev: the cfg4 model's conv2 weight gradient under the library / knobs of this process -> gpurun_out/conv2_grad_<tag>.pt,
plus the two operands of that weight-gradient call"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from kgcn_amd import BatchedAdjacency, BatchedCSR, data_util as D, models, ops
from kgcn_amd import _lib
tag = sys.argv[1]
dev = torch.device("cuda:0")
B, N, F, TASKS = 4096, 50, 81, 12
sizes, g, r, c, rng = bench.gen_tox21_like(B, N, seed=4)
chan = D.normalize_adj(D.FlatAdjacency(g, r, c, np.ones(g.shape[0], np.float32), B, N))
adj = BatchedAdjacency([BatchedCSR.from_arrays(chan.graph, chan.row, chan.col, chan.val, B, N, N, device=dev)])
valid = np.arange(N)[None, :] < sizes[:, None]
x = torch.from_numpy(rng.standard_normal((B, N, F)).astype(np.float32) * valid[:, :, None]).to(dev)
labels = torch.from_numpy((rng.random((B, TASKS)) < 0.3).astype(np.float32)).to(dev)
mask_label = torch.from_numpy((rng.random((B, TASKS)) < 0.8).astype(np.float32)).to(dev)
mask = torch.from_numpy((rng.random(B) < 0.95).astype(np.float32)).to(dev)
en = torch.from_numpy(sizes.astype(np.int32)).to(dev)
# record the operands of every wide weight-gradient call
calls = []
orig_w = ops._Dense._wgrad
def spy(ctx, x2d, w, gy, yact, m, din, dout, need_w, need_b, fuse_dact):
    dw, db = orig_w(ctx, x2d, w, gy, yact, m, din, dout, need_w, need_b, fuse_dact)
    if din == 256 and dout == 256 and dw is not None:
        ref = torch.zeros((din, dout), dtype=torch.float64, device=gy.device)
        for s0 in range(0, m, 16384):
            ref += x2d[s0:s0 + 16384].double().t() @ gy[s0:s0 + 16384].double()
        e = (dw.double() - ref).abs()
        colmax = gy.abs().amax(0); colmed = gy.abs().median(0).values
        xcolmax = x2d.abs().amax(0)
        calls.append((m, float(e.max() / ref.abs().max()), float(ref.abs().max())))
        worst_cols = torch.argsort(e.amax(0), descending=True)[:6]
        print("wgrad %dx%d m=%d: rel err %.2e; worst dy columns %s  their |dy| max %s median %s ; |dy| global max %.3e  nan %d inf %d"
              % (din, dout, m, e.max() / ref.abs().max(), worst_cols.tolist(), colmax[worst_cols].tolist(), colmed[worst_cols].tolist(),
                 float(gy.abs().max()), int(torch.isnan(gy).sum()), int(torch.isinf(gy).sum())))
        best_cols = torch.argsort(e.amax(0))[:4]
        print("   best columns %s max %s median %s; x col max range %.3g..%.3g; zero rows of dy: %d, zero rows of x: %d"
              % (best_cols.tolist(), colmax[best_cols].tolist(), colmed[best_cols].tolist(), float(xcolmax.min()), float(xcolmax.max()),
                 int((gy.abs().amax(1) == 0).sum()), int((x2d.abs().amax(1) == 0).sum())))
        torch.save({"x": x2d.detach().cpu(), "dy": gy.detach().cpu()}, os.path.join(ROOT, "gpurun_out", "wg_ops_%s_%d.pt" % (tag, len(calls)))) if tag == "new" and len(calls) == 1 else None
    return dw, db
ops._Dense._wgrad = staticmethod(spy)
torch.manual_seed(0)
model = models.MultitaskGCN(1, TASKS, ragged=False).to(dev)
model(x, adj, enabled_node_nums=en)
gen = torch.Generator(device="cpu").manual_seed(1)
with torch.no_grad():
    for p in model.parameters():
        if p.dim() == 1 or p.shape[0] == 1:
            p.add_(torch.randn(p.shape, generator=gen).to(p.device) * 0.1)
calls.clear()
logits = model(x, adj, enabled_node_nums=en)
cost, cost_sum = models.masked_sigmoid_ce(logits, labels, mask, mask_label, 2.0)
cost.backward()
torch.cuda.synchronize()
print(calls)
out = {"grad": model.conv2.w[0].grad.detach().cpu()}
torch.save(out, os.path.join(ROOT, "gpurun_out", "conv2_grad_%s.pt" % tag))
