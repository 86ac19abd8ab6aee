#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_dense_edges.py tests/test_gpu_model.py tests/test_gpu_bench_size.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for rep in 1 2 3; do for c in cfg4; do python bench.py --config $c --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],4))"; done; done
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $OUT/st -o b -- python $REPO/bench.py --profile --config cfg4 --steps 20 --warmup 5 > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r05k/st/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:45]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), r["AverageNs"][:9].rjust(10), r["TotalDurationNs"].rjust(12))
PY
rm -rf $OUT/st
