#!/bin/bash
# round 3, after the pinned prefetch of the table GEMM: full GPU suite, smoke, headline line, cfg4 / cfg5 / cfg3 profiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/w
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/w/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/w/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/w/bench.json 2> gpurun_out/w/bench.err; echo "bench rc=$?"
tools/profile_config.sh r03n_cfg5 20 3 --config cfg5 > /dev/null 2>&1
tools/profile_config.sh r03n_cfg4 20 3 --config cfg4 > /dev/null 2>&1
tools/profile_config.sh r03n_cfg3 50 5 --config cfg3 > /dev/null 2>&1
for t in cfg5 cfg4 cfg3; do sed -n 2,8p gpurun_out/prof_r03n_$t/summary.txt | cut -c1-140; grep GHz gpurun_out/prof_r03n_$t/summary.txt | head -2; done
