#!/bin/bash
# A/B of the ragged aggregation: block kernel (LDS tile per row block of whole molecules) vs the row-chunk kernel, cfg4
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_spmm_blocks.py tests/test_gpu_ragged.py -m gpu -x -q 2>&1 | tail -2
python tools/spmm_block_bench.py --d 256,50 2>&1 | grep "^[0-9]" | cut -c1-150
for v in 1 0 1 0; do
  KGCN_SPMM_BLOCKS=$v timeout 300 python bench.py --config cfg4 --steps 40 --warmup 8 --no-cpu-baseline --profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg4 blocks=$v', d['ms_per_step'])"
done
