#!/bin/bash
# round 4: where the time of the f16 GEMM goes -- variant libraries + rocprofv3 counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" prev gh1 gh2 gh3 gh4; do
  if [ -z "$v" ]; then timeout 120 python tools/gemmh_prof.py 2>/dev/null; else KGCN_HIP_LIB=$PWD/build/variants/libkgcn_$v.so timeout 120 python tools/gemmh_prof.py 2>/dev/null; fi
done | tee gpurun_out/r04b_variants.jsonl
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04b_prof; rm -rf $OUT; mkdir -p $OUT
pass() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" -d $OUT/$tag -o g -- python $R/tools/gemmh_prof.py 117888 256 256 10 > $OUT/$tag.log 2>&1; }
pass stats --stats
pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
pass pmc_inst --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS
pass pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
for d in stats pmc_sq pmc_inst pmc_tcc pmc_fetch pmc_write pmc_grbm; do
  find $OUT/$d -name '*_kernel_stats.csv' -exec cp {} $OUT/${d}_kernel_stats.csv \; 2>/dev/null
  find $OUT/$d -name '*_counter_collection.csv' -exec cp {} $OUT/${d}_counters.csv \; 2>/dev/null
  rm -rf $OUT/$d
done
ls -la $OUT; head -20 $OUT/stats_kernel_stats.csv
python - <<'P'
import csv, collections, glob, os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04b_prof"
for f in sorted(glob.glob(out+"/pmc_*_counters.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:60]; agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
    print("==",os.path.basename(f))
    for k in agg:
        if "gemm" in k or "reduce" in k:
            print(k, {c: "%.4g"%(v/n[(k,c)]) for c,v in agg[k].items()})
P
