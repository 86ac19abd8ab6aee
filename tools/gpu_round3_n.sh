#!/bin/bash
# round 3, final profiles: the headline pair (cfg2, refreshes profiles/traffic_cfg2.json) and every model configuration
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/profile_round.sh r03h > /dev/null 2>&1
tools/profile_config.sh r03h_cfg4 20 3 --config cfg4 > /dev/null 2>&1
tools/profile_config.sh r03h_cfg5 20 3 --config cfg5 > /dev/null 2>&1
tools/profile_config.sh r03h_cfg3 50 5 --config cfg3 > /dev/null 2>&1
tools/profile_config.sh r03h_cfg1_b30 200 10 --config cfg1 > /dev/null 2>&1
tools/profile_config.sh r03h_cfg1_b4096 100 5 --config cfg1 --graphs 20000 --batch 4096 > /dev/null 2>&1
tail -6 gpurun_out/prof_r03h/summary.txt
for t in cfg4 cfg5 cfg3 cfg1_b30 cfg1_b4096; do sed -n 2,3p gpurun_out/prof_r03h_$t/summary.txt | cut -c1-160; done
python tools/stack_sweep.py > gpurun_out/r03h_stack_sweep.json 2>/dev/null
