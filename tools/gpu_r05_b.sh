#!/bin/bash
# round 5, call b: first run of the one-pass dense backward (gemmb.hip): its tests, then the microbenchmark
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05b
timeout 600 python -m pytest tests/test_gpu_dense_bwd.py -x -q > gpurun_out/r05b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05b/pytest.log
tail -25 gpurun_out/r05b/pytest.log
timeout 300 python tools/dense_bwd_bench.py > gpurun_out/r05b/bench.jsonl 2>&1; cat gpurun_out/r05b/bench.jsonl | tail -8
