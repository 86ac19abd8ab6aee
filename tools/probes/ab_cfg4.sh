cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  timeout 300 python bench.py --config cfg4 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bf16-split', d['ms_per_step'])"
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_WGRADX=f32 timeout 300 python bench.py --config cfg4 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('f32-mfma  ', d['ms_per_step'])"
done
