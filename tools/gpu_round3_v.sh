#!/bin/bash
# round 3, FINAL: full GPU suite, smoke, the headline line, and rocprofv3 summaries of every model configuration at HEAD
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/v
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/v/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/v/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/v/bench.json 2> gpurun_out/v/bench.err; echo "bench rc=$?"
tools/profile_config.sh r03m_cfg4 20 3 --config cfg4 > /dev/null 2>&1
tools/profile_config.sh r03m_cfg5 20 3 --config cfg5 > /dev/null 2>&1
tools/profile_config.sh r03m_cfg3 50 5 --config cfg3 > /dev/null 2>&1
tools/profile_config.sh r03m_cfg1_b30 200 10 --config cfg1 > /dev/null 2>&1
for t in cfg4 cfg5 cfg3 cfg1_b30; do sed -n 2,3p gpurun_out/prof_r03m_$t/summary.txt | cut -c1-140; done
python tools/train_bench.py > gpurun_out/v/train_bench.json 2>/dev/null; tail -c 600 gpurun_out/v/train_bench.json
