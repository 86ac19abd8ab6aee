#!/bin/bash
# round 3, call r: 32-column slices of the tile aggregation kernel for 17..32-node graphs -- GPU suite, headline profile (refreshes
# profiles/traffic_cfg2.json: spmm.hip is one of the hashed sources), bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r/tests.log
tools/profile_round.sh r03j > /dev/null 2>&1
tail -12 gpurun_out/prof_r03j/summary.txt
timeout 300 python bench.py > gpurun_out/r/bench.json 2> gpurun_out/r/bench.err; echo "bench rc=$?"
