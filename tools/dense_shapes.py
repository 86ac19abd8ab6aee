import sys, torch, json
sys.path.insert(0, '/root/repo')
from kgcn_amd._lib import lib, ptr, current_stream, check
dev = torch.device('cuda:0')
def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a,b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b) for a,b in ev)
    return ts[len(ts)//2]
for (m, din, dout) in [(4096*50, 256, 256), (4096*50, 81, 256), (1_000_000, 256, 256), (3_200_000, 64, 64), (1_000_000, 128, 128)]:
    x = torch.randn(m, din, device=dev); w = torch.randn(din, dout, device=dev)*0.05; b = torch.randn(dout, device=dev)
    y = torch.empty(m, dout, device=dev); dy = torch.randn(m, dout, device=dev)
    dw = torch.empty(din, dout, device=dev); db = torch.empty(dout, device=dev)
    t1 = timeit(lambda: check(lib.kgcn_dense_fwd_f32(ptr(x), m, din, din, ptr(w), dout, 0, ptr(b), ptr(y), dout, dout, current_stream())))
    dx = torch.empty(m, din, device=dev)
    t2 = timeit(lambda: check(lib.kgcn_dense_fwd_f32(ptr(dy), m, dout, dout, ptr(w), dout, 1, 0, ptr(dx), din, din, current_stream())))
    wsb = lib.kgcn_dense_wgrad_workspace_bytes(m, din, dout); ws = torch.empty(wsb//4, device=dev)
    t3 = timeit(lambda: check(lib.kgcn_dense_wgrad_f32(ptr(x), din, ptr(dy), dout, m, din, dout, ptr(dw), ptr(db), ptr(ws), wsb, current_stream())))
    fl = 2.0*m*din*dout
    tt = timeit(lambda: torch.matmul(x, w))
    by = 4.0*m*(din+dout)
    print(json.dumps({"m":m,"din":din,"dout":dout,"fwd_ms":round(t1,3),"fwd_TF":round(fl/t1/1e9,1),"fwd_GBs_alg":round(by/t1/1e6),
      "dx_ms":round(t2,3),"dx_TF":round(fl/t2/1e9,1),"wgrad_ms":round(t3,3),"wgrad_TF":round(fl/t3/1e9,1),"torch_mm_ms":round(tt,3),"torch_TF":round(fl/tt/1e9,1)}))
