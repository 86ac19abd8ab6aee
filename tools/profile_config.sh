#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for a bench.py configuration (cfg4 / cfg5 / cfg3 / cfg2).
#   1. un-profiled bench line                                   -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (+ --profile: no per-call roofline pass, no CPU baseline)
#   3. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides the kernel trace)
#   4. tools/profile_config_summary.py: per-kernel table of the TIMED steps -> gpurun_out/prof_<tag>/summary.txt
# usage: tools/profile_config.sh <tag> <steps> <warmup> [bench args...]     (copy summary.txt into profiles/)
set -u
TAG=$1; STEPS=$2; WARM=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --steps $STEPS --warmup $WARM "$@" > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $REPO/bench.py --profile --steps $STEPS --warmup $WARM $*"
pass() { tag=$1; shift; timeout 900 rocprofv3 --kernel-trace "$@" -d $OUT/$tag -o bench -- $BENCH > $OUT/$tag.log 2>&1; }
pass stats --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS
python $REPO/tools/profile_config_summary.py $OUT $STEPS > $OUT/summary.txt 2>&1
# the rocpd databases are tens of MB each (gpurun copies back <= 64 MiB): keep the summary, the logs and the stats csv only
for d in stats pmc_fetch pmc_write pmc_grbm pmc_sq pmc_inst; do
  find $OUT/$d -name '*_kernel_stats.csv' -exec cp {} $OUT/${d}_kernel_stats.csv \; 2>/dev/null
  rm -rf $OUT/$d
done
cat $OUT/summary.txt
