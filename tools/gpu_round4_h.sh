#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/build/variants/libkgcn_dev.so
KGCN_HIP_LIB=$L KGCN_GEMMH=0 timeout 300 python tools/gemmh_debug2.py old 2>&1 | tail -2
KGCN_HIP_LIB=$L KGCN_GEMMH=w timeout 300 python tools/gemmh_debug2.py new 2>&1 | tail -2
exit 0
python - <<'P'
import torch, numpy as np
a=torch.load("gpurun_out/conv2_grad_old.pt")["grad"].numpy().astype(np.float64); b=torch.load("gpurun_out/conv2_grad_new.pt")["grad"].numpy().astype(np.float64)
e=np.abs(a-b); print("max|a|",np.abs(a).max(),"max err",e.max(), "mean err", e.mean())
print("err by row block of 32:", e.reshape(8,32,256).max(axis=(1,2)))
print("err by col block of 32:", e.reshape(256,8,32).max(axis=(0,2)))
r,c=np.unravel_index(e.argmax(), e.shape); print("worst at", r, c, a[r,c], b[r,c])
print("rows with err>1e-6:", np.nonzero(e.max(1)>1e-6)[0][:40]); print("cols with err>1e-6:", np.nonzero(e.max(0)>1e-6)[0][:40])
P
