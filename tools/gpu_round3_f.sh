mkdir -p gpurun_out/g6
python -m pytest tests -x -q -m gpu > gpurun_out/g6/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g6/pytest_gpu.log
tail -5 gpurun_out/g6/pytest_gpu.log
bash tools/profile_config.sh r03c_cfg3 50 5 --config cfg3 > gpurun_out/g6/prof_cfg3.log 2>&1
bash tools/profile_config.sh r03c_cfg4 20 3 --config cfg4 > gpurun_out/g6/prof_cfg4.log 2>&1
bash tools/profile_config.sh r03c_cfg5 20 3 --config cfg5 > gpurun_out/g6/prof_cfg5.log 2>&1
for n in cfg3 cfg4 cfg5; do head -3 gpurun_out/prof_r03c_$n/summary.txt | cut -c1-160; done
python tools/train_bench.py 30 4096 > gpurun_out/g6/train_bench.json 2>/dev/null
python tools/stack_sweep.py 30 512 4096 > gpurun_out/g6/stack_sweep.json 2>/dev/null
python tools/ragged_spmm_bench.py > gpurun_out/g6/ragged_spmm.json 2>/dev/null
python tools/pack_time.py > gpurun_out/g6/pack_time.txt 2>&1
