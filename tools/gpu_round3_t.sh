#!/bin/bash
# round 3, call t: bf16 register-split weight gradient of the narrow-input layer (wgradxb) -- A/B against the f32-MFMA kernel, parity, cfg4
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/t
for r in 1 2; do
KGCN_WX_CHECK=1 timeout 200 python tools/wgradx_bench.py 2>&1 | tail -2
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_WGRADX=f32 timeout 200 python tools/wgradx_bench.py 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dense_edges.py tests/test_gpu_ragged.py tests/test_gpu_bench_size.py tests/test_gpu_large_sizes.py -x -q -m gpu > gpurun_out/t/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t/tests.log
timeout 300 python bench.py --config cfg4 > gpurun_out/t/cfg4.json 2>/dev/null
KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_WGRADX=f32 timeout 300 python bench.py --config cfg4 > gpurun_out/t/cfg4_f32.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/t/cfg4.json','gpurun_out/t/cfg4_f32.json'):
    d=json.loads(open(f).read().strip().split('\n')[-1]); print(f, d['value'], d['ms_per_step'])
"
