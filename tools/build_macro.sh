#!/bin/bash
# build arcnerf_amd/lib/alt_<tag>.so from the working tree with extra compiler flags on ONE source file (other objects as built):
#   tools/build_macro.sh <tag> <file-without-.hip> "<-DX=1 ...>"
TAG=$1; FILE=$2; DEFS=$3
set -e
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics $DEFS -c arcnerf_amd/csrc/$FILE.hip -o $T/$FILE.o
OBJS=""
for o in volume batch bitfield hashgrid encode mlp gemm render neus step_glue optim api; do
  if [ "$o" == "$FILE" ]; then OBJS="$OBJS $T/$FILE.o"; else OBJS="$OBJS arcnerf_amd/_build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o arcnerf_amd/lib/alt_$TAG.so $OBJS
rm -rf $T
ls -la arcnerf_amd/lib/alt_$TAG.so
