#!/bin/bash
# development: forward time of the tile route with phases switched off (KGCN_DEV_KNOBS build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for abl in 0 1 2 4 8 16 3 7 15 31; do
  echo "abl=$abl: $(KGCN_S2_ABL=$abl timeout 120 python tools/stack_kernel_bench.py 4096 2>&1 | grep -A2 route2 | tr -d '\n ')"
done
