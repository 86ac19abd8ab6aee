#!/bin/bash
# round 3, call q: full GPU suite + profiles of the model configurations after the launch-shape change and the stale-row fix
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/q
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/q/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/q/tests.log
tools/profile_config.sh r03i_cfg3 50 5 --config cfg3 > /dev/null 2>&1
tools/profile_config.sh r03i_cfg5 20 3 --config cfg5 > /dev/null 2>&1
tools/profile_config.sh r03i_cfg4 20 3 --config cfg4 > /dev/null 2>&1
for t in cfg3 cfg5 cfg4; do sed -n 2,3p gpurun_out/prof_r03i_$t/summary.txt | cut -c1-160; done
timeout 300 python bench.py > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; echo "bench rc=$?"
