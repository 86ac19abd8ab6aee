#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06i; mkdir -p $OUT
for v in gb_head gb_hot; do echo "== $v"; KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$v.so timeout 600 python tools/dense_bwd_bench.py 200000 2>/dev/null | cut -c1-220; done
for v in gb_head gh_hot; do echo "== $v (gemmh family)"; KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$v.so timeout 600 python tools/gemmh_bench.py --rows 200000 --quick 2>/dev/null | tail -12 | cut -c1-250; done
