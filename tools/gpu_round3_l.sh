python -m pytest tests/test_gpu_model.py tests/test_gpu_ragged.py -x -q -m gpu 2>&1 | tail -3
python tools/stack_sweep.py 30 256 512 1024 2>/dev/null
for b in 30 4096; do python bench.py --config cfg1 --batch $b --steps 200 --warmup 10 --profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1 batch $b', round(d['ms_per_step'],4), round(d['value']))"; done
