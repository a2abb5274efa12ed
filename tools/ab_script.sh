#!/bin/bash
# alternate arcnerf_amd/lib/alt_old.so and alt_new.so under one script in one session: tools/ab_script.sh "<python script + args>" [reps]
CMD=$1; REPS=${2:-3}
cd arcnerf_amd/lib; cp libarcnerf_hip.so keep.so; cd ../..
for rep in $(seq $REPS); do
  for v in old new; do
    cp arcnerf_amd/lib/alt_$v.so arcnerf_amd/lib/libarcnerf_hip.so
    echo -n "$v: "; python $CMD 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
cp arcnerf_amd/lib/keep.so arcnerf_amd/lib/libarcnerf_hip.so; rm arcnerf_amd/lib/keep.so
