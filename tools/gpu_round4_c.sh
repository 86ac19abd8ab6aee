#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" prev; do
  for m in 117888 200000; do
  if [ -z "$v" ]; then timeout 120 python tools/gemmh_prof.py $m 2>/dev/null; else KGCN_HIP_LIB=$PWD/build/variants/libkgcn_$v.so timeout 120 python tools/gemmh_prof.py $m 2>/dev/null; fi
  done
done | tee gpurun_out/r04c_variants.jsonl
timeout 300 python tools/gemmh_bench.py --rows 117888 --shapes 256x256,84x256 2>&1 >/dev/null | tail -3
timeout 900 python -m pytest tests/test_gpu_dense_edges.py -x -q 2>&1 | tail -5
