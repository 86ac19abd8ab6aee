#!/bin/bash
# Runs on the GPU box (via gpurun): the judged profile of bench.py at HEAD.
#   1. un-profiled bench line                                   -> gpurun_out/prof_<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats                         -> stats/
#   3. PMC passes, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass; GRBM for the clock)
#   4. tools/profile_summary.py: summary text + traffic json    -> gpurun_out/prof_<tag>/{summary.txt,traffic_cfg2.json}
# usage: tools/profile_round.sh <tag> [bench args...]     (copy the summaries you want judged into profiles/)
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --steps 50 --warmup 5 $* > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $REPO/bench.py --no-cpu-baseline --steps 30 --warmup 5 $*"
pass() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace "$@" -d $OUT/$tag -o bench -- $BENCH > $OUT/$tag.log 2>&1; }
pass stats --stats
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
python $REPO/tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
