#!/bin/bash
# round 5, call q: the dot form of the one-pass backward (GIN's d epsilon inside gemmb.hip): W' k-steps kept in registers x the k-step at
# which the dot operand is requested; parity of the default, cfg5 bench lines of the variants alternating on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dense_bwd.py tests/test_gpu_dense_edges.py tests/test_gpu_parity.py -m gpu -q -x -k "gin or dot or epsilon or dense_bwd" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2 3; do
for v in old w8k5 w8k0 w9k0 w8k2; do
  KGCN_HIP_LIB=$REPO/build/variants/libkgcn_dot_$v.so python bench.py --config cfg5 --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4))"
done
done
