#!/bin/bash
# round 5, call s: gemmh_fwd_kernel's two-phase schedule (wave groups half a tile apart) against the lock-step form
# (build/variants/libkgcn_lockstep.so: -DGH_TWO_PHASES=0): parity, the probe, cfg5 / cfg4 bench lines alternating on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_parity.py tests/test_gpu_bench_size.py tests/test_gpu_large_sizes.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/gemmh_fwd_probe.py 200000 2>/dev/null | tail -8
for rep in 1 2 3; do
for v in lockstep new; do
  if [ $v = new ]; then unset KGCN_HIP_LIB; else export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$v.so; fi
  for c in cfg5 cfg4; do python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['ms_per_step'],4))"; done
done
done
