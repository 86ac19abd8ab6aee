#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/gemmh_bench.py --rows 117888,200000,117900 --shapes 256x256,128x256,200x192 2>&1 >/dev/null | grep -v amdgpu.ids | sed 's/fwd_err.*wgrad_err/ wgrad_err/' | tail -9
timeout 600 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_bench_size.py -x -q 2>&1 | tail -4
timeout 120 python tools/gemmh_prof.py 117888 2>/dev/null
