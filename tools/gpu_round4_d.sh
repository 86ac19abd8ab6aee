#!/bin/bash
# rocprofv3 counters of the wide-layer GEMM microbenchmark (tools/gemmh_prof.py): stats + PMC passes, summarised on the box
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r04d_prof}; rm -rf $OUT; mkdir -p $OUT
shift
pass() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" -d $OUT/$tag -o g -- python $R/tools/gemmh_prof.py ${ARGS:-117888 256 256 10} > $OUT/$tag.log 2>&1; }
pass stats --stats
pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
pass pmc_tcp --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
python $R/tools/rocpd_kernels.py $OUT 10 > $OUT/summary.txt 2>&1
for d in stats pmc_sq pmc_inst pmc_tcc pmc_fetch pmc_write pmc_grbm pmc_tcp; do rm -rf $OUT/$d; done
grep -i "gemm\|wgrad\|==\|reduce" $OUT/summary.txt
tail -3 $OUT/pmc_tcp.log
