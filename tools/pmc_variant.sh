#!/bin/bash
# PMC passes of tools/variant_bench.py for one library variant (GPU box).  usage: tools/pmc_variant.sh <name>
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
name=$1
OUT=$REPO/gpurun_out/pmc_$name
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$name.so
[ -f "$KGCN_HIP_LIB" ] || unset KGCN_HIP_LIB
export VB_TIMING_ONLY=1
CMD="python $REPO/tools/variant_bench.py"
pass() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag -o p -- $CMD > $OUT/$tag.log 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE
pass sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
python $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
