#!/bin/bash
# Run ON THE GPU BOX: per-kernel stats (rocprofv3 --kernel-trace --stats) of the default bench step with each prebuilt library
# arcnerf_amd/lib/alt_<tag>.so in turn: tools/prof_alts.sh tagA tagB ...   -> gpurun_out/prof_alts/<tag>_top.txt
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_alts
mkdir -p $OUT
cp arcnerf_amd/lib/libarcnerf_hip.so arcnerf_amd/lib/keep.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $ROOT/arcnerf_amd/lib/alt_$v.so $ROOT/arcnerf_amd/lib/libarcnerf_hip.so
  rm -rf /tmp/pa_$v
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pa_$v -o $v --output-format csv -- python $ROOT/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-other-configs --no-psnr > $OUT/$v.json 2> $OUT/$v.log
  find /tmp/pa_$v -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${v}_kernel_stats.csv
  python - $OUT/${v}_kernel_stats.csv $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print('==', sys.argv[2])
for r in rows[:12]:
    print('%-60s calls %5s avg %8.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cp $ROOT/arcnerf_amd/lib/keep.so $ROOT/arcnerf_amd/lib/libarcnerf_hip.so; rm $ROOT/arcnerf_amd/lib/keep.so
