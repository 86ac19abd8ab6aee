#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or bf16_split or cfg2_full_size or graphconv" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'bwd', r['frac'], r['launch_ms']['median_ms'], 'fwd', r['fwd_kernel']['frac'], r['fwd_kernel']['launch_ms']['median_ms'], r.get('frac_of_probe_rate'), r['box']['sustained']['sclk_mhz'], r['box']['hbm_probe']['add_2r1w'])
PY
