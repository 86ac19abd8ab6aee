#!/bin/bash
# round 3, call p: full GPU suite after the launch-shape change + stale-row fix of the gathered-gradient GEMM
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/p
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/p/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/p/tests.log
timeout 300 python bench.py > gpurun_out/p/bench.json 2> gpurun_out/p/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/p/bench.json
