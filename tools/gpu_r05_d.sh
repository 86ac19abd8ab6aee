#!/bin/bash
# round 5, call d: the model-level suites with the one-pass dense backward in the route, then the bench lines of every config
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_dense_edges.py tests/test_gpu_model.py tests/test_gpu_dense_bwd.py tests/test_gpu_configs.py tests/test_gpu_large_sizes.py tests/test_gpu_step_abi.py -x -q > gpurun_out/r05d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05d/pytest.log
tail -15 gpurun_out/r05d/pytest.log
bash tools/bench_lines.sh r05d > gpurun_out/r05d/lines.txt 2>&1; cat gpurun_out/r05d/lines.txt
