mkdir -p gpurun_out/g7
python -m pytest tests/test_gpu_dense_edges.py -x -q -m gpu -k "wgrad_with_fused" 2>&1 | tail -4
for i in 1 2; do
for c in "cfg4" "cfg4 --no-wgrad-dact" "cfg4 --contract-first"; do
  n=$(echo $c | tr -d ' -')_$i
  python bench.py --config $c --steps 30 --warmup 3 --graphs 40000 --profile > gpurun_out/g7/$n.json 2> gpurun_out/g7/$n.err
  python -c "
import json;d=json.loads(open('gpurun_out/g7/$n.json').read().strip().splitlines()[-1]);print('$n', d['ms_per_step'], d['value'])"
done
done
