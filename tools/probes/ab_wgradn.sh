cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dense_edges.py tests/test_gpu_large_sizes.py -x -q -m gpu -k "dense" 2>&1 | tail -2
for r in 1 2; do
for v in new lds; do
  if [ $v = new ]; then E=""; else E="KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_WGRADN=lds"; fi
  env $E timeout 300 python bench.py --config cfg4 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=[t for t in d['roofline']['per_call_table'] if t['entry']=='kgcn_dense_wgrad_f32' and '256x50' in t['shape']]
print('$v', d['ms_per_step'], r[0]['us'] if r else None, r[0]['frac_hbm'] if r else None)
"
done
done
