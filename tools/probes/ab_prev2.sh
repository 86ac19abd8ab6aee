cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in cfg5 cfg4; do
for r in 1 2; do
  timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg new ', d['ms_per_step'])"
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_prev.so timeout 300 python bench.py --config $cfg --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg prev', d['ms_per_step'])"
done
done
