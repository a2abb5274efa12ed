#!/bin/bash
# Run ON THE GPU BOX: GPU-busy time per step (sum of kernel durations, rocprofv3 --kernel-trace --stats) against the step time of a
# module-path config: tools/gpu_busy.sh <config> [steps=8] [warmup=3]
CFG=$1; STEPS=${2:-8}; WARM=${3:-3}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gb_$CFG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/gb_$CFG -o gb --output-format csv -- python $ROOT/bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline > /tmp/gb_$CFG.json 2>/dev/null
python - $CFG $STEPS $WARM <<'PY'
import csv, glob, json, sys
cfg, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob('/tmp/gb_%s/**/*kernel_stats.csv' % cfg, recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = sum(int(r['Calls']) for r in rows)
d = json.load(open('/tmp/gb_%s.json' % cfg))
print('%s: kernel time %.2f ms per step, %d launches per step, step under rocprof %.2f ms' % (cfg, tot / 1e6 / (steps + warm), n // (steps + warm), d['ms_per_step']))
PY
