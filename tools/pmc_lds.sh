#!/bin/bash
# LDS bank conflicts per kernel of the bench step (counters only): conflict cycles / LDS active cycles
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lds_1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/lds_1 -o c --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /tmp/lds_1.log 2>&1
tail -3 /tmp/lds_1.log | cut -c1-300
python - <<'PY'
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/lds_1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'arcn::' in r['Kernel_Name']:
            vals[r['Kernel_Name'].split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
def med(v):
    v = sorted(v); return v[len(v)//2] if v else 0.0
print('%-44s %8s %7s %12s %12s %9s %10s' % ('kernel', 'waves', 'LDS/w', 'idx_active', 'bank_confl', 'confl%', 'lds/dur%'))
for k, c in sorted(vals.items(), key=lambda kv: -med(kv[1]['GRBM_GUI_ACTIVE'])):
    w = med(c['SQ_WAVES']) or 1
    ia, bc = med(c['SQ_LDS_IDX_ACTIVE']), med(c['SQ_LDS_BANK_CONFLICT'])
    dur = med(c['GRBM_GUI_ACTIVE']) / 8.0
    print('%-44s %8d %7.0f %12.0f %12.0f %9.1f %10.1f' % (k, w, med(c['SQ_INSTS_LDS']) / w, ia, bc, 100 * bc / max(ia, 1), 100 * ia / 256.0 / max(dur, 1)))
PY
