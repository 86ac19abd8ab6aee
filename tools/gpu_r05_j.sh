#!/bin/bash
# round 5, call j: small fusions of cfg4 (augmented rows assembled in place, act' inside the narrow dX kernel, BN's d gamma / d beta in the
# step's one reduction launch): the tests that cover them, bench lines, rocprofv3 of cfg4 / cfg3
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_dense_edges.py tests/test_gpu_model.py tests/test_gpu_bench_size.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for rep in 1 2 3; do for c in cfg4 cfg3; do python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4))"; done; done
bash tools/profile_config.sh r05j_cfg4 20 5 --config cfg4 > /dev/null 2>&1; head -60 gpurun_out/prof_r05j_cfg4/summary.txt | cut -c1-125
