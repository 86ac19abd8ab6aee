mkdir -p gpurun_out/g9
python -m pytest tests -x -q -m gpu > gpurun_out/g9/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g9/pytest_gpu.log
tail -4 gpurun_out/g9/pytest_gpu.log
for i in 1 2; do
for c in "cfg4" "cfg4 --no-side-wgrad" "cfg5" "cfg5 --no-side-wgrad" "cfg3" "cfg3 --no-side-wgrad"; do
  n=$(echo $c | tr -d ' -')_$i
  python bench.py --config $c --steps 30 --warmup 3 --profile $( [[ $c == cfg4* ]] && echo "--graphs 40000" ) > gpurun_out/g9/$n.json 2> gpurun_out/g9/$n.err
  python -c "
import json;d=json.loads(open('gpurun_out/g9/$n.json').read().strip().splitlines()[-1]);print('$n', round(d['ms_per_step'],4), round(d['value']))" || tail -3 gpurun_out/g9/$n.err
done
done
