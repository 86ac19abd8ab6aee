#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "" gh5 gh6 gh7; do
  if [ -z "$v" ]; then timeout 120 python tools/gemmh_prof.py 117888 2>/dev/null; else KGCN_HIP_LIB=$PWD/build/variants/libkgcn_$v.so timeout 120 python tools/gemmh_prof.py 117888 2>/dev/null; fi
done | tee gpurun_out/r04k_variants.jsonl
