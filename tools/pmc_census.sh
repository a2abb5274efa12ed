#!/bin/bash
# instruction census of every kernel of the bench step (two counter passes, counters only)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cen_1 /tmp/cen_2
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/cen_1 -o c --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d /tmp/cen_2 -o c --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/cen_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'arcn::' in r['Kernel_Name']:
            vals[r['Kernel_Name'].split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
def med(v):
    v = sorted(v); return v[len(v)//2] if v else 0.0
print('%-44s %8s %7s %7s %6s %6s %8s %9s %7s' % ('kernel', 'waves', 'VALU/w', 'SALU/w', 'LDS/w', 'MFMA/w', 'dur_us', 'VALUbusy%', 'wait%'))
for k, c in sorted(vals.items(), key=lambda kv: -med(kv[1]['GRBM_GUI_ACTIVE'])):
    w = med(c['SQ_WAVES']) or 1
    dur_cyc = med(c['GRBM_GUI_ACTIVE']) / 8.0          # summed over 8 XCDs
    valu = med(c['SQ_INSTS_VALU']); mf = med(c['SQ_INSTS_MFMA'])
    # VALU issue cycles per SIMD: 4 cycles per wave64 VALU op, 32 per f32 16x16x4 MFMA
    busy = ((valu - mf) * 4 + mf * 32) / 1024.0
    print('%-44s %8d %7.0f %7.0f %6.0f %6.0f %8.1f %9.0f %7.0f' % (k, w, valu / w, med(c['SQ_INSTS_SALU']) / w, med(c['SQ_INSTS_LDS']) / w, mf / w,
          dur_cyc / 2400.0, 100 * busy / max(dur_cyc, 1), 100 * med(c['SQ_WAIT_INST_ANY']) / max(med(c['SQ_WAVE_CYCLES']), 1)))
PY
