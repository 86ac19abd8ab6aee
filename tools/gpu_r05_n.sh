#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dense_bwd.py -m gpu -q -x -k "reads_out_inside" > $OUT/pytest1.log 2>&1; tail -4 $OUT/pytest1.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
  for c in cfg5; do
    (cd $REPO/build/ab/prev && python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | line prev $c)
    (cd $REPO && python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | line new $c)
  done
done
