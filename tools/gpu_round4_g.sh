#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=tests/test_gpu_bench_size.py::test_multitask_model_at_batch_4096_ragged_equals_padded
for k in 0 f d w fd; do
  echo "== KGCN_GEMMH=$k"
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_GEMMH=$k timeout 300 python -m pytest $T -x -q 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
