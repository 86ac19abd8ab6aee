#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "" gh2 gh3 gh4; do
  if [ -z "$v" ]; then timeout 120 python tools/gemmh_prof.py 117888 2>/dev/null; else KGCN_HIP_LIB=$PWD/build/variants/libkgcn_$v.so timeout 120 python tools/gemmh_prof.py 117888 2>/dev/null; fi
done | tee gpurun_out/r04e_variants.jsonl
bash tools/gpu_round4_d.sh r04e_prof
