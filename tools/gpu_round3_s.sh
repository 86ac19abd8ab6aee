#!/bin/bash
# round 3, call s: soak + fuzz after the last kernel changes (gemm3 launch shapes / gathered-gradient fix, SpMM slices)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/s
for spec in "cfg5 1000" "cfg3 5000" "cfg4 1000" "cfg1 5000"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline > gpurun_out/s/soak_$1.json 2> gpurun_out/s/soak_$1.err; echo "$1 rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/s/soak_$1.json').read().strip().split('\n')[-1]); print('$1', d['steps'], d['ms_per_step'], d['value'], {k: d['config'].get(k) for k in ('loss_first','loss_last') if k in d['config']})
"
done
timeout 300 python bench.py --steps 3000 --warmup 5 --no-cpu-baseline > gpurun_out/s/soak_cfg2.json 2>/dev/null; python -c "
import json
d=json.loads(open('gpurun_out/s/soak_cfg2.json').read().strip().split('\n')[-1]); print('cfg2', d['steps'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['spmm_kernel']['forward']['frac'])
"
for s in 21 22 23; do timeout 600 python tools/fuzz_gpu.py 400 $s 2>&1 | tail -2; done
