#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in cfg4 cfg5 cfg3; do
  bash tools/profile_config.sh r04a_$cfg 20 5 --config $cfg > /dev/null 2>&1
  head -60 gpurun_out/prof_r04a_$cfg/summary.txt | cut -c1-200 | head -45
done
timeout 300 python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('cfg4', d['ms_per_step'], {k:r[k] for k in r if k not in ('per_call_table','method','peak_note')}); [print('   ', t['entry'], t['shape'], t['calls'], t['us'], t['bound'], t['frac'], t['frac_hbm'], t['frac_mfma']) for t in r['per_call_table'][:12]]"
timeout 300 python bench.py --config cfg5 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('cfg5', d['ms_per_step'], {k:r[k] for k in r if k not in ('per_call_table','method','peak_note')}); [print('   ', t['entry'], t['shape'], t['calls'], t['us'], t['bound'], t['frac'], t['frac_hbm'], t['frac_mfma']) for t in r['per_call_table'][:8]]"
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
