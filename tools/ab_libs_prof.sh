# per-kernel averages (rocprofv3 --kernel-trace --stats) of the bench step for alt_old.so / alt_new.so; MATCH = substrings of the kernel names to print
#   MATCH="march_count hashgrid_fwd" bash tools/ab_libs_prof.sh
ROOT=$(pwd)
MATCH=${MATCH:-"mlp_ scatter hashgrid_fwd march_count"}
cd arcnerf_amd/lib; cp libarcnerf_hip.so keep.so
for v in old new; do
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
cp keep.so libarcnerf_hip.so; rm keep.so
