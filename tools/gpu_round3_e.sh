mkdir -p gpurun_out/g5
python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/g5/pytest_model.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g5/pytest_model.log
tail -25 gpurun_out/g5/pytest_model.log
python tools/train_bench.py 30 4096 20000 > gpurun_out/g5/train_bench.json 2> gpurun_out/g5/train_bench.err; tail -3 gpurun_out/g5/train_bench.err
python -c "
import json; d=json.load(open('gpurun_out/g5/train_bench.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, {kk: round(vv['ms_per_step'],4) for kk,vv in v.items()})
"
python tools/pack_time.py > gpurun_out/g5/pack_time.txt 2>&1; tail -12 gpurun_out/g5/pack_time.txt
