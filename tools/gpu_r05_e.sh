#!/bin/bash
# round 5, call e: the headline experiments VERDICT r04 item 2 asked to be MEASURED (time box: 40 GPU-minutes)
#  (a) SpMM: the 32-column slices of a graph as waves of one workgroup, CSR staged once (spmm_slices_kernel) against the
#      one-workgroup-per-slice form (build/variants/libkgcn_noslicewaves.so): bench lines alternating + PMC traffic / instruction mix
#  (b) graphconv_bwd_planes_kernel non-persistent: k workgroups per CU slot (build/variants/libkgcn_dev.so, KGCN_BWD_GRID_MULT)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r05e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_sizes.py -x -q -k "spmm or bconv or gin or graphconv or batched or op_wrappers" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
line() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]; s=r.get("spmm_kernel",{})
print(sys.argv[1].split("/")[-1], "value %.1f M graphs/s"%(d["value"]/1e6), "ms %.4f"%d["ms_per_step"], "bwd frac", round(r.get("frac",0),4),
      "| spmm fwd", s.get("forward",{}).get("frac"), s.get("forward",{}).get("us_median"), "adjoint", s.get("adjoint",{}).get("frac"), s.get("adjoint",{}).get("us_median"))
PY
}
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/a_slicewaves_$rep.json; line $OUT/a_slicewaves_$rep.json
  KGCN_HIP_LIB=$REPO/build/variants/libkgcn_noslicewaves.so python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/a_perslice_$rep.json; line $OUT/a_perslice_$rep.json
done
for k in 1 2 4 8; do
  KGCN_HIP_LIB=$REPO/build/variants/libkgcn_dev.so KGCN_BWD_GRID_MULT=$k python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/b_gridmult_$k.json; line $OUT/b_gridmult_$k.json
done
cd /tmp && export TMPDIR=/tmp
for v in slicewaves perslice; do
  [ $v = perslice ] && export KGCN_HIP_LIB=$REPO/build/variants/libkgcn_noslicewaves.so
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${v}_$tag -o p -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $OUT/pmc_${v}_$tag.log 2>&1
  done
  unset KGCN_HIP_LIB
done
python - <<'PY'
import csv,glob,os,collections
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r05e"
for d in sorted(glob.glob(out+"/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"].split("(")[0][:60]
            if "spmm" in k or "planes" in k: agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,cs in agg.items():
            print(os.path.basename(d), k, {c:(round(sum(v)/len(v)),len(v)) for c,v in cs.items()})
PY
