#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -3
