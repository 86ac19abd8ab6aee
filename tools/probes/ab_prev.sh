# A/B of the shipped library against build/variants/libkgcn_prev.so: table GEMM microbenchmark + whole steps
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/gemm_cut_bench.py > gpurun_out/ab_new.json 2>/dev/null
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_prev.so timeout 300 python tools/gemm_cut_bench.py > gpurun_out/ab_prev.json 2>/dev/null
python - <<'P'
import json
a=json.load(open("gpurun_out/ab_new.json")); b=json.load(open("gpurun_out/ab_prev.json"))
for m in a:
    print("%7s rows  fwd %7.1f us (prev %7.1f)  dx_dact %7.1f (%7.1f)  err %.1e %.1e" % (m, a[m]["fwd_us"], b[m]["fwd_us"], a[m]["dx_dact_us"], b[m]["dx_dact_us"], a[m]["fwd_err"], a[m]["dx_err"]))
P
for cfg in cfg5 cfg4; do
for r in 1 2; do
  timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg new ', d['ms_per_step'])"
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_prev.so timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg prev', d['ms_per_step'])"
done
done
