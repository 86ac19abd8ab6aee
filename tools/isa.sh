#!/bin/bash
# tools/isa.sh <file.hip>: compile one translation unit of kgcn_amd/csrc with -save-temps into /tmp/kgcn_isa/<name>/ and print the
# register / spill / LDS figures of its kernels (the ISA is in <name>-hip-amdgcn-amd-amdhsa-gfx950.s there)
set -e
name=$(basename "$1" .hip)
out=/tmp/kgcn_isa/$name
mkdir -p "$out"
cd "$(dirname "$0")/../kgcn_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast ${EXTRA} \
  -c "$name.hip" -o "$out/$name.o" -save-temps=obj 2> "$out/err.txt" || { grep -v "hip-link" "$out/err.txt" | head -30; echo "COMPILE FAILED"; exit 1; }
grep -v "hip-link" "$out/err.txt" | grep -v "^$" | head -20 || true
grep -E "\.(vgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size):|^\s+\.name:" "$out/$name-hip-amdgcn-amd-amdhsa-gfx950.s" \
  | sed 's/\s\+/ /g' | paste - - - - - | sed 's/ - //'
