import sys, torch
sys.path.insert(0, '/root/repo')
from kgcn_amd import ops
torch.manual_seed(0)
dev = torch.device('cuda:0')
bad = 0
for (M, din, dout) in [(2000, 256, 256), (2000, 64, 64), (777, 81, 256), (300, 50, 50), (3200000, 64, 64)]:
    x = torch.randn(M, din, device=dev); g = torch.randn(M, dout, device=dev)
    w = torch.randn(din, dout, device=dev, requires_grad=True); b = torch.zeros(dout, device=dev, requires_grad=True)
    ref = x.double().t() @ g.double()
    for it in range(40 if M < 100000 else 6):
        w.grad = None; b.grad = None
        xx = x.clone().requires_grad_(True)
        y = ops.dense(xx, w, b)
        y.backward(g)
        err = float((w.grad.double() - ref).abs().max())
        errx = float((xx.grad.double() - g.double() @ w.detach().double().t()).abs().max())
        if err > 1e-2 * float(ref.abs().max()) or errx > 1e-2:
            bad += 1
            d = (w.grad.double() - ref).abs()
            print("BAD", M, din, dout, it, err, errx, "rows with err:", (d.max(1).values > 1e-2).nonzero().flatten()[:10].tolist(), "cols:", (d.max(0).values > 1e-2).nonzero().flatten()[:10].tolist())
print("bad runs:", bad)
