#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06k; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_bconv_loop.py tests/test_gpu_bench.py tests/test_gpu_step_no_aten.py -m gpu -q > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log | cut -c1-250
