# alternate two prebuilt libraries (arcnerf_amd/lib/alt_old.so, alt_new.so) in one session; prints step and per-entry kernel times
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in 1 2 3; do
  for v in old new; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$v', round(d['ms_per_step'],4), 'mlp_fwd', round(k['mlp_fwd'],4), 'mlp_bwd', round(k['mlp_bwd'],4))")
  done
done
cp keep.so libarcnerf_hip.so; rm keep.so
