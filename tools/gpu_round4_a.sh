#!/bin/bash
# round 4, first GPU call: the f16 two-piece GEMM family -- correctness + timing against the round-3 library, edge-value suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/gemmh_bench.py --rows 117888,200000 --shapes 256x256,84x256,128x256 > gpurun_out/r04a_gemmh_new.json 2> gpurun_out/r04a_gemmh_new.log
echo "new rc=$?"; cat gpurun_out/r04a_gemmh_new.log | tail -12
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_prev.so timeout 300 python tools/gemmh_bench.py --rows 117888,200000 --shapes 256x256 --quick > gpurun_out/r04a_gemmh_prev.json 2> gpurun_out/r04a_gemmh_prev.log
echo "prev rc=$?"; tail -4 gpurun_out/r04a_gemmh_prev.log
timeout 900 python -m pytest tests/test_gpu_dense_edges.py -x -q 2>&1 | tail -15
