#!/bin/bash
# round 5, call g: forward GEMM with the next tile's staging laid between its MFMAs (gemmh_fwd_kernel<0,16>) against the round-4 form
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_bench_size.py tests/test_gpu_dense_bwd.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for rep in 1 2; do for v in interleave nointerleave; do
  [ $v = nointerleave ] && export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_nointerleave.so
  for c in cfg4 cfg5; do python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $OUT/${v}_${c}_$rep.json
    python -c "
import json; d=json.loads(open('$OUT/${v}_${c}_$rep.json').read().strip().split('\n')[-1]); print('$v $c #$rep', round(d['ms_per_step'],4), 'ms')"; done
  unset KGCN_HIP_LIB; done; done
