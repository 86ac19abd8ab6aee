#!/bin/bash
# round 5, call i: rows of a stage staged by the dW waves (gemmb.hip GB_DW_ROWS = 0 / 1 / 2 / 4): parity of the default, A/B of the variants
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dense_bwd.py tests/test_gpu_dense_edges.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
for v in 0 1 2 4; do
  export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_dw$v.so
  echo "== dw rows $v rep $rep" >> $OUT/ab.txt
  python tools/dense_bwd_bench.py 200000 >> $OUT/ab.txt 2>&1
  for c in cfg5 cfg4; do python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4))" >> $OUT/ab.txt; done
done
done
cat $OUT/ab.txt
