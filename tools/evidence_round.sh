#!/bin/bash
# The judged evidence of round 4 at HEAD: bench line + rocprofv3 kernel trace + PMC traffic for every configuration.
#   cfg2 (headline): tools/profile_round.sh -> gpurun_out/prof_r04v/{summary.txt, traffic_cfg2.json, bench.json}
#   cfg4 / cfg5 / cfg3 / cfg1: tools/profile_config.sh -> gpurun_out/prof_r04v_<cfg>/{summary.txt, traffic.json, bench.json}
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh r04v > /dev/null 2>&1
for cfg in cfg4 cfg5 cfg3 cfg1; do
  bash tools/profile_config.sh r04v_$cfg 20 5 --config $cfg > /dev/null 2>&1
done
for d in gpurun_out/prof_r04v gpurun_out/prof_r04v_cfg*; do echo "== $d"; head -4 $d/summary.txt | cut -c1-160; python -c "
import json,sys; d=json.loads(open('$d/bench.json').read().strip().split('\n')[-1]); print(d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('bound'))"; done
