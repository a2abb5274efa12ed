#!/bin/bash
# alternate prebuilt libraries arcnerf_amd/lib/alt_<tag>.so in one session (three rounds): tools/ab_alts.sh tagA tagB [tagC ...]
# prints the step, its median, and the in-step times of the gather / scatter / nets
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$v', round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), {n: round(k[n],4) for n in ('hashgrid_fwd','hashgrid_bwd','mlp_fwd','mlp_bwd') if n in k}, 'gather alone', round(d['roofline_lookup'].get('alone_launch_ms',0),4))")
  done
done
cp keep.so libarcnerf_hip.so; rm keep.so
