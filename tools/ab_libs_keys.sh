# alternate two prebuilt libraries (arcnerf_amd/lib/alt_old.so, alt_new.so) in one session; prints the step time and the kernel_ms entries named in $KEYS
#   KEYS="march_count hashgrid_fwd" REPS=3 bash tools/ab_libs_keys.sh
KEYS=${KEYS:-"march_count hashgrid_fwd"}
REPS=${REPS:-3}
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in $(seq 1 $REPS); do
  for v in old new; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | KEYS="$KEYS" python -c "
import sys,json,os
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('$v', 'step', round(d['ms_per_step'],4), ' '.join('{} {}'.format(n, round(k.get(n, float('nan')),4)) for n in os.environ['KEYS'].split()), 'lookup_frac', round(d.get('roofline_lookup',{}).get('frac',0),3))")
  done
done
cp keep.so libarcnerf_hip.so; rm keep.so
