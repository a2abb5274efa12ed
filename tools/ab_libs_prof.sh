# per-kernel averages (rocprofv3 --stats) of the MLP kernels for alt_old.so / alt_new.so
ROOT=$(pwd)
cd arcnerf_amd/lib; cp libarcnerf_hip.so keep.so
for v in old new; do
  cp alt_$v.so libarcnerf_hip.so
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abp_$v && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/abp_$v -o p --output-format csv -- python $ROOT/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-configs > /dev/null 2>&1)
  echo "== $v"; python - <<PY
import csv,glob
f=glob.glob('/tmp/abp_$v/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'mlp_' in r['Name'] or 'scatter' in r['Name'] or 'hashgrid_fwd' in r['Name']: print(r['Name'][11:60].ljust(50), r['Calls'], r['AverageNs'][:8])
PY
done
cp keep.so libarcnerf_hip.so; rm keep.so
