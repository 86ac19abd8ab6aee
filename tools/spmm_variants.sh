#!/bin/bash
# development: libkgcn_hip.so variants whose spmm.hip is compiled with extra defines (what each part of spmm_block_kernel costs)
# into build/variants/libkgcn_<tag>.so.   usage: tools/spmm_variants.sh tag1:"-DSPB_VARIANT=1" tag2:"-DSPB_ITEMS_N=2" ...
set -e
cd "$(dirname "$0")/../kgcn_amd/csrc"
mkdir -p ../../build/variants
OBJS=$(ls *.o | grep -v '^spmm.o$')
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $defs -c spmm.hip -o /tmp/spmm_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libkgcn_$tag.so $OBJS /tmp/spmm_$tag.o
done
ls ../../build/variants
