#!/bin/bash
# round 5, call r: the random sweeps at HEAD (tools/fuzz_gemmh.py: the f16 two-piece GEMMs incl. the one-pass backward through ops.dense;
# tools/fuzz_gpu.py: every kernel family against the oracle)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05r; mkdir -p $OUT
for seed in 11 12 13; do timeout 900 python tools/fuzz_gemmh.py 40 $seed 2>/dev/null | tail -42 > $OUT/fuzz_gemmh_$seed.txt; tail -1 $OUT/fuzz_gemmh_$seed.txt; grep -c FAIL $OUT/fuzz_gemmh_$seed.txt; done
for seed in 21 22 23; do timeout 1200 python tools/fuzz_gpu.py 40 $seed 2>/dev/null | tail -15 > $OUT/fuzz_gpu_$seed.txt; tail -3 $OUT/fuzz_gpu_$seed.txt; done
