#!/bin/bash
# step times of every configuration at HEAD (no per-call roofline pass, no CPU baseline)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in cfg4 cfg5 cfg3 cfg1 cfg2; do
  for rep in 1 2; do
  timeout 300 python bench.py --config $cfg --steps 40 --warmup 8 --no-cpu-baseline --profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', d['ms_per_step'], d['value'], d['unit'])"
  done
done
