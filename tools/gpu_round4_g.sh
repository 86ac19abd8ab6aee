#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=tests/test_gpu_bench_size.py::test_cfg5_model_at_20000_graphs_against_the_oracle
for k in "KGCN_GEMMH=0" "KGCN_GEMMH=fdw KGCN_WGRADL=0" "KGCN_GEMMH=fd" "KGCN_GEMMH=w" "KGCN_GEMMH=fw" "KGCN_GEMMH=dw"; do
  echo "== $k"
  env KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so $k timeout 300 python -m pytest $T -x -q 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
