#!/bin/bash
# GPU box, repo root: tools/prof_config.sh <tag> <config> [extra bench args]: rocprofv3 --kernel-trace --stats of one module-path config; the
# summary goes to gpurun_out/<tag>_<config>_kernel_stats.csv (copy into profiles/ to keep it)
set -u
TAG=$1; CFG=$2; shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python bench.py --config $CFG --steps 32 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/${TAG}_bench_$CFG.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG --output-format csv -- \
    python $ROOT/bench.py --config $CFG --steps 16 --warmup 4 --no-cpu-baseline "$@" > $OUT/${TAG}_bench_${CFG}_under_rocprof.json 2> $OUT/${TAG}_trace.log
cd $ROOT
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_${CFG}_kernel_stats.csv
python - <<PY
import csv, json
rows = list(csv.DictReader(open('$OUT/${TAG}_${CFG}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
arcn = sum(float(r['TotalDurationNs']) for r in rows if 'arcn' in r['Name'])
print('kernel time total %.2f ms in the profiled run, arcn share %.1f %%' % (tot / 1e6, 100 * arcn / tot))
for r in rows[:28]:
    print('%6.2f%% %7d calls %9.1f us avg  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:110]))
print(open('$OUT/${TAG}_bench_$CFG.json').read()[:300])
PY
