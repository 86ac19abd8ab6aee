#!/bin/bash
# Builds variants of libkgcn_hip.so that differ in -D flags of ONE source (VSRC, default fused) -- HERE, cross-compiled:
#   [VSRC=gemm3] tools/variants.sh build name1 "-DFLAG=1" name2 "-DFLAG=2 -DOTHER" ...
# and times them on the GPU box:  [VBENCH=narrow_probe] tools/variants.sh run name1 name2 ...   (-> gpurun_out/variants/<name>.json)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/kgcn_amd/csrc
mode=$1; shift
VSRC=${VSRC:-fused}
SLP=""; [ "$VSRC" = fused ] && SLP="-fno-slp-vectorize"
mkdir -p $REPO/build/variants $REPO/gpurun_out/variants
if [ "$mode" = build ]; then
  while [ $# -gt 0 ]; do
    name=$1; flags=$2; shift 2
    objs=""
    for o in misc ragged train stack stack_tile skinny spmm dense gemm3 gemmh gemmb wtable gemmn wgradn wgradx narrow fused pack coopack gat bn; do
      if [ $o = $VSRC ]; then objs="$objs $REPO/build/variants/${VSRC}_$name.o"; else objs="$objs $CS/$o.o"; fi
    done
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $SLP $flags \
        -c $CS/$VSRC.hip -o $REPO/build/variants/${VSRC}_$name.o 2> $REPO/build/variants/$name.log &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/build/variants/libkgcn_$name.so $objs &&
      echo "built $name" || { echo "FAILED $name"; tail -5 $REPO/build/variants/$name.log; } ) &
  done
  wait
else
  for name in "$@"; do
    KGCN_HIP_LIB=$REPO/build/variants/libkgcn_$name.so timeout 300 python $REPO/tools/${VBENCH:-variant_bench}.py \
      > $REPO/gpurun_out/variants/$name.json 2> $REPO/gpurun_out/variants/$name.err || echo "$name failed"
    echo "$name: $(cat $REPO/gpurun_out/variants/$name.json)"
  done
fi
