#!/bin/bash
# development: build libkgcn_hip.so variants whose gemmh.hip is compiled with -DGH_VARIANT=n (what each part of the f16 GEMM
# costs) into build/variants/libkgcn_gh<n>.so; the other objects are the shipped ones
set -e
cd "$(dirname "$0")/../kgcn_amd/csrc"
mkdir -p ../../build/variants
OBJS=$(ls *.o | grep -v '^gemmh.o$')
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -DGH_VARIANT=$v -c gemmh.hip -o /tmp/gemmh_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libkgcn_gh$v.so $OBJS /tmp/gemmh_v$v.o
done
ls -la ../../build/variants
