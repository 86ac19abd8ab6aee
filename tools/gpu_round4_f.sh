#!/bin/bash
# whole GPU suite + the configuration steps with the shipped library
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for cfg in cfg4 cfg5 cfg3 cfg1; do
  for r in 1 2; do
    timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', d['ms_per_step'], d['value'])"
  done
done
timeout 300 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
