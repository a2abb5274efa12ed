# tools/ab_libs_env.sh "ENV=.." ...: run the alt_new.so library under several environments, alt_old.so with defaults as the baseline
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in 1 2; do
  cp alt_old.so libarcnerf_hip.so
  (cd ../..; python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('old', round(d['ms_per_step'],4), 'mlp_fwd', round(k['mlp_fwd'],4), 'mlp_bwd', round(k['mlp_bwd'],4))")
  cp alt_new.so libarcnerf_hip.so
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    (cd ../..; env $e python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('new [$cfg]', round(d['ms_per_step'],4), 'mlp_fwd', round(k['mlp_fwd'],4), 'mlp_bwd', round(k['mlp_bwd'],4))")
  done
done
cp keep.so libarcnerf_hip.so; rm keep.so
