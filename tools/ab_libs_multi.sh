# alternate N prebuilt libraries (arcnerf_amd/lib/alt_<v>.so for v in $VARIANTS) in one session: step time + kernel_ms entries of $KEYS, then
# (PROF=1) per-kernel averages under rocprofv3 for the kernels matching $MATCH
#   VARIANTS="base slim5" KEYS="hashgrid_bwd" REPS=3 PROF=1 MATCH="scatter" bash tools/ab_libs_multi.sh
VARIANTS=${VARIANTS:-"base new"}
KEYS=${KEYS:-"hashgrid_bwd"}
REPS=${REPS:-3}
MATCH=${MATCH:-"scatter"}
ROOT=$(pwd)
cd arcnerf_amd/lib
cp libarcnerf_hip.so keep.so
for rep in $(seq 1 $REPS); do
  for v in $VARIANTS; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | KEYS="$KEYS" python -c "
import sys,json,os
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('$v'.ljust(8), 'step', round(d['ms_per_step'],4), ' '.join('{} {}'.format(n, round(k.get(n, float('nan')),4)) for n in os.environ['KEYS'].split()), 'roofline', round(d['roofline']['frac'],4))")
  done
done
if [ "${PROF:-0}" == "1" ]; then
for v in $VARIANTS; do
  cp alt_$v.so libarcnerf_hip.so
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abp_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abp_$v -o p --output-format csv -- python $ROOT/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-configs --no-psnr > /dev/null 2>&1)
  echo "== $v"; MATCH="$MATCH" python - <<PY
import csv,glob,os
f=glob.glob('/tmp/abp_$v/**/*kernel_stats.csv', recursive=True)[0]
keys=os.environ['MATCH'].split()
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in keys): print(r['Name'][:70].ljust(72), r['Calls'], r['AverageNs'][:9])
PY
done
fi
cp keep.so libarcnerf_hip.so; rm keep.so
