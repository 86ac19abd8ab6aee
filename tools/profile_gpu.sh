#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{stats,pmc_*}  (copy the summaries you want judged into profiles/)
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 3 $*"
# 1) kernel trace + stats (no counters in this pass)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- $BENCH > $OUT/stats.log 2>&1
# 2) PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_inst -o bench -- $BENCH > $OUT/pmc_inst.log 2>&1
find $OUT -name '*.csv' | head -50
