#!/bin/bash
# same-box A/B: build/ab/prev (a copy of the previous commit, its own library) against the working tree; cfg4 / cfg3 bench lines alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_bench.py tests/test_gpu_ragged.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do
  for c in cfg4 cfg3; do
    (cd $REPO/build/ab/prev && python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | line prev $c)
    (cd $REPO && python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | line new $c)
  done
done
