mkdir -p gpurun_out/g10
bash tools/profile_config.sh r03d_cfg3 50 5 --config cfg3 > gpurun_out/g10/prof_cfg3.log 2>&1
bash tools/profile_config.sh r03d_cfg4 20 3 --config cfg4 > gpurun_out/g10/prof_cfg4.log 2>&1
bash tools/profile_config.sh r03d_cfg5 20 3 --config cfg5 > gpurun_out/g10/prof_cfg5.log 2>&1
for n in cfg3 cfg4 cfg5; do head -3 gpurun_out/prof_r03d_$n/summary.txt | cut -c1-160; done
python tools/train_bench.py 30 4096 > gpurun_out/g10/train_bench.json 2>/dev/null
python examples/train_synthetic.py > gpurun_out/g10/train_synthetic.log 2>&1; tail -5 gpurun_out/g10/train_synthetic.log
