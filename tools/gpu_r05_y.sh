#!/bin/bash
# round 5, call y: longer random sweeps at HEAD
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05y; mkdir -p $OUT
for seed in 31 32 33 34; do timeout 1500 python tools/fuzz_gpu.py 150 $seed 2>/dev/null | tail -12 > $OUT/fuzz_gpu_$seed.txt; tail -2 $OUT/fuzz_gpu_$seed.txt; done
for seed in 41 42; do timeout 1500 python tools/fuzz_gemmh.py 120 $seed 2>/dev/null > $OUT/fuzz_gemmh_$seed.txt; tail -1 $OUT/fuzz_gemmh_$seed.txt; grep -c FAIL $OUT/fuzz_gemmh_$seed.txt; done
