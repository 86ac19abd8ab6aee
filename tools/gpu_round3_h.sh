mkdir -p gpurun_out/g8
python -m pytest tests -x -q -m gpu > gpurun_out/g8/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g8/pytest_gpu.log
tail -6 gpurun_out/g8/pytest_gpu.log
for c in cfg5 cfg4 cfg3; do
  python bench.py --config $c --steps 30 --warmup 3 > gpurun_out/g8/$c.json 2> gpurun_out/g8/$c.err
  python -c "
import json;d=json.loads(open('gpurun_out/g8/$c.json').read().strip().splitlines()[-1]);print('$c', d['ms_per_step'], d['value'])
for r in d['roofline'].get('per_call_table',[])[:14]: print('   %-30s %-44s x%d %8.1f us hbm %.3f mfma %.3f'%(r['entry'],r['shape'],r['calls'],r['us'],r['frac_hbm'],r['frac_mfma_f32']))"
done
