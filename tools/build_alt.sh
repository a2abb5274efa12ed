#!/bin/bash
# build arcnerf_amd/lib/alt_<tag>.so from the hashgrid.hip of a git revision (other objects as built): tools/build_alt.sh <rev> <tag> [file=hashgrid]
REV=$1; TAG=$2; FILE=${3:-hashgrid}
set -e
T=$(mktemp -d)
cp arcnerf_amd/csrc/*.hpp $T/
mkdir -p $T/../include_dummy
git show $REV:arcnerf_amd/csrc/$FILE.hip > $T/$FILE.hip
INC=$(pwd)/include
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -I$INC -I$(pwd)/arcnerf_amd/csrc -c $T/$FILE.hip -o $T/$FILE.o
OBJS=""
for o in volume batch bitfield hashgrid encode mlp gemm render neus step_glue optim api; do
  if [ "$o" == "$FILE" ]; then OBJS="$OBJS $T/$FILE.o"; else OBJS="$OBJS arcnerf_amd/_build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o arcnerf_amd/lib/alt_$TAG.so $OBJS
rm -rf $T
ls -la arcnerf_amd/lib/alt_$TAG.so
