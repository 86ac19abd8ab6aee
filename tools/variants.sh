#!/bin/bash
# Builds variants of libkgcn_hip.so that differ in -D flags of fused.hip (HERE, cross-compiled) -- usage:
#   tools/variants.sh build name1 "-DFLAG=1" name2 "-DFLAG=2 -DOTHER" ...
# and times them on the GPU box:  tools/variants.sh run name1 name2 ...   (-> gpurun_out/variants/<name>.json)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/kgcn_amd/csrc
mode=$1; shift
mkdir -p $REPO/build/variants $REPO/gpurun_out/variants
if [ "$mode" = build ]; then
  while [ $# -gt 0 ]; do
    name=$1; flags=$2; shift 2
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize $flags \
        -c $CS/fused.hip -o $REPO/build/variants/fused_$name.o 2> $REPO/build/variants/$name.log &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/build/variants/libkgcn_$name.so \
        $CS/misc.o $CS/spmm.o $CS/dense.o $CS/gemm3.o $REPO/build/variants/fused_$name.o $CS/pack.o $CS/gat.o $CS/bn.o &&
      echo "built $name" || { echo "FAILED $name"; tail -5 $REPO/build/variants/$name.log; } ) &
  done
  wait
else
  for name in "$@"; do
    KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$name.so timeout 300 python $REPO/tools/variant_bench.py \
      > $REPO/gpurun_out/variants/$name.json 2> $REPO/gpurun_out/variants/$name.err || echo "$name failed"
    echo "$name: $(cat $REPO/gpurun_out/variants/$name.json)"
  done
fi
