#!/bin/bash
# round 3, call u: register-split weight gradients (wgradxb, wgradnb) shipped -- full GPU suite, cfg4 profile, bench lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/u
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/u/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/u/tests.log
tools/profile_config.sh r03l_cfg4 20 3 --config cfg4 > /dev/null 2>&1
sed -n 2,14p gpurun_out/prof_r03l_cfg4/summary.txt | cut -c1-150
timeout 300 python bench.py > gpurun_out/u/bench.json 2> gpurun_out/u/bench.err; echo "bench rc=$?"
