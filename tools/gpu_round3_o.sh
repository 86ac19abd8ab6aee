#!/bin/bash
# round 3, call o: column-cut launch shape of the table GEMM -- parity tests, A/B against whole blocks, cfg3 / cfg5 steps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/o
echo "== HEAD library on the new gather case (stale broadcast row beyond the first tile of a workgroup?)"
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_head.so timeout 300 python -m pytest tests/test_gpu_dense_edges.py -x -q -m gpu -k "gather and 3616" 2>&1 | tail -4
echo "== this tree"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dense_edges.py -x -q -m gpu -k "dense or activation" > gpurun_out/o/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/o/tests.log
timeout 300 python tools/gemm_cut_bench.py > gpurun_out/o/cut_on.json 2> gpurun_out/o/cut_on.err; echo "cut_on rc=$?"
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_GEMM3_CUT=0 timeout 300 python tools/gemm_cut_bench.py > gpurun_out/o/cut_off.json 2> gpurun_out/o/cut_off.err; echo "cut_off rc=$?"
python - <<'P'
import json
a=json.load(open("gpurun_out/o/cut_on.json")); b=json.load(open("gpurun_out/o/cut_off.json"))
for m in a:
    print("%7s rows  fwd %7.1f us (whole blocks %7.1f)  dx_dact %7.1f (%7.1f)  err %.1e %.1e %.1e" % (m, a[m]["fwd_us"], b[m]["fwd_us"], a[m]["dx_dact_us"], b[m]["dx_dact_us"], a[m]["fwd_err"], a[m]["dx_err"], a[m]["dpre_err"]))
P
for c in cfg3 cfg5; do
  timeout 300 python bench.py --config $c > gpurun_out/o/$c.json 2> gpurun_out/o/$c.err; echo "$c rc=$?"
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_GEMM3_CUT=0 timeout 300 python bench.py --config $c > gpurun_out/o/${c}_off.json 2> gpurun_out/o/${c}_off.err
  python -c "
import json
for f in ('gpurun_out/o/$c.json','gpurun_out/o/${c}_off.json'):
    d=json.loads(open(f).read().strip().split('\n')[-1]); print(f, d['value'], d['ms_per_step'])
"
done
