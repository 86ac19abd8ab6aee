#!/bin/bash
# round 3, last profiles of the round: every model configuration with the committed library
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/profile_config.sh r03f_cfg4 20 3 --config cfg4 > /dev/null 2>&1
tools/profile_config.sh r03f_cfg5 20 3 --config cfg5 > /dev/null 2>&1
tools/profile_config.sh r03f_cfg3 50 5 --config cfg3 > /dev/null 2>&1
tools/profile_config.sh r03f_cfg1_b30 200 10 --config cfg1 > /dev/null 2>&1
tools/profile_config.sh r03f_cfg1_b4096 100 5 --config cfg1 --graphs 20000 --batch 4096 > /dev/null 2>&1
for t in cfg4 cfg5 cfg3 cfg1_b30 cfg1_b4096; do head -3 gpurun_out/prof_r03f_$t/summary.txt | cut -c1-200; done
python tools/stack_sweep.py > gpurun_out/r03f_stack_sweep.json 2>/dev/null
