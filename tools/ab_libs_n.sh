# alternate several prebuilt libraries (arcnerf_amd/lib/alt_<tag>.so) in one session: tools/ab_libs_n.sh REPS tag1 tag2 ...
REPS=$1; shift
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in $(seq $REPS); do
  for v in "$@"; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --steps 192 --warmup 16 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$v', round(d['ms_per_step'],4), 'scatter', round(d['roofline']['avg_launch_ms'],4), 'gather', round(k['hashgrid_fwd'],4), 'mlp_fwd', round(k['mlp_fwd'],4), 'mlp_bwd', round(k['mlp_bwd'],4))")
  done
done
cp keep.so libarcnerf_hip.so; rm keep.so
