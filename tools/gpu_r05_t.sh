#!/bin/bash
# which waves share a SIMD: the two-phase groups of gemmh_fwd_kernel by bit 2 (shipped), bit 1, bit 0 of the wave index; cfg5 lines on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2 3; do
for v in lockstep new group1 group0; do
  if [ $v = new ]; then unset KGCN_HIP_LIB; else export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$v.so; fi
  python bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4))"
done
done
