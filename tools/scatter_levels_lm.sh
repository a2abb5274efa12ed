#!/bin/bash
# producer / consumer time of the binned scatter per level (kernel trace only); run on the GPU box from the repo root
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for m in 0xffff 0x1 0x2 0x10 0x20 0x40 0x80 0x100 0x200 0x400 0x800 0x1000 0x2000 0x4000 0x8000; do
  rm -rf /tmp/sl
  ARCN_SCATTER_LEVELS=$m timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/sl -o sl --output-format csv -- python $ROOT/tools/exp_scatter_lm.py 10 > /tmp/sl.log 2>&1
  python - "$m" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/sl/**/*kernel_stats.csv', recursive=True):
    out = []
    for r in csv.DictReader(open(f)):
        if 'scatter' in r['Name'] or 'memset' in r['Name'].lower():
            out.append('%s %.1f us x%s' % (r['Name'].split('(')[0].split('::')[-1][:28], float(r['AverageNs']) / 1e3, r['Calls']))
    print('levels', sys.argv[1], ' | '.join(out))
PY
done
