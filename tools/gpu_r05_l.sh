#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_model.py tests/test_gpu_bench_size.py tests/test_gpu_bench.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for rep in 1 2 3; do for c in cfg4; do python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4))"; done; done
