#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <script.py> <kernel-substring> "<COUNTERS pass 1>" ["<COUNTERS pass 2>" ...]
# prints the per-launch median of every counter for the launches of the matching kernel (counters only, no other trace domain)
SCRIPT=$1; KERN=$2; shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 300 rocprofv3 --pmc $SET -d /tmp/pk_$i -o pk --output-format csv -- python $ROOT/$SCRIPT > /dev/null 2>&1
done
python - "$KERN" <<'PY'
import csv, glob, sys
from collections import defaultdict
kern = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/pk_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            vals[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in vals.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print('    %-32s median %14.1f  (n=%d)' % (c, v[len(v) // 2], len(v)))
PY
