cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests -x -q -m gpu -k "ragged or model_multitask or batch_4096" 2>&1 | tail -2
for r in 1 2; do
  timeout 300 python bench.py --config cfg4 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=[t for t in d['roofline']['per_call_table'] if 'ragged_gather_fwd' in t['entry']]
print('cfg4', d['ms_per_step'], [(t['us'], t['frac_hbm']) for t in r])
"
done
