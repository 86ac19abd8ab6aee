#!/bin/bash
# whole GPU suite + the configuration steps, shipped library against the round-3 one
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in cfg4 cfg5 cfg3; do
  for r in 1 2; do
    timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg new ', d['ms_per_step'])"
    KGCN_HIP_LIB=$PWD/build/variants/libkgcn_prev.so timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg prev', d['ms_per_step'])"
  done
done
