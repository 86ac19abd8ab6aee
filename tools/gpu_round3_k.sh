mkdir -p gpurun_out/g11
bash tools/profile_config.sh r03e_cfg1_b30 200 10 --config cfg1 > gpurun_out/g11/p1.log 2>&1
bash tools/profile_config.sh r03e_cfg1_b4096 100 10 --config cfg1 --batch 4096 > gpurun_out/g11/p2.log 2>&1
bash tools/profile_config.sh r03e_cfg3 50 5 --config cfg3 > gpurun_out/g11/p3.log 2>&1
bash tools/profile_config.sh r03e_cfg4 20 3 --config cfg4 > gpurun_out/g11/p4.log 2>&1
bash tools/profile_config.sh r03e_cfg5 20 3 --config cfg5 > gpurun_out/g11/p5.log 2>&1
for n in cfg1_b30 cfg1_b4096 cfg3 cfg4 cfg5; do head -3 gpurun_out/prof_r03e_$n/summary.txt | cut -c1-140; done
