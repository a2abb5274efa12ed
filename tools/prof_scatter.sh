#!/bin/bash
# usage: tools/prof_scatter.sh <tag>   (env is inherited): kernel split of tools/exp_scatter.py under rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc_$1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sc_$1 -o sc --output-format csv -- python /root/repo/tools/exp_scatter.py 2>&1 | grep samples
python - <<PY
import csv,glob
f=glob.glob('/tmp/sc_$1/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    if 'scatter' in r['Name'] or 'fillBuffer' in r['Name']:
        print('   ', r['Name'][11:40], r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
PY
