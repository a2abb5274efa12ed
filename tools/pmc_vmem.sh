#!/bin/bash
# global-memory request efficiency per kernel of the bench step (counters only): L2 requests per vector memory instruction
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vm_1 /tmp/vm_2
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/vm_1 -o c --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /tmp/vm_1.log 2>&1
timeout 400 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE -d /tmp/vm_2 -o c --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /tmp/vm_2.log 2>&1
tail -2 /tmp/vm_1.log | cut -c1-200; tail -2 /tmp/vm_2.log | cut -c1-200
python - <<'PY'
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/vm_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'arcn::' in r['Kernel_Name']:
            vals[r['Kernel_Name'].split('(')[0][-44:]][r['Counter_Name']].append(float(r['Counter_Value']))
def med(v):
    v = sorted(v); return v[len(v)//2] if v else 0.0
print('%-44s %8s %9s %9s %10s %10s %8s %8s %10s %8s' % ('kernel', 'waves', 'rd_inst/w', 'wr_inst/w', 'rd_req', 'wr_req', 'req/rd', 'req/wr', 'tcp_acc', 'TA_busy%'))
for k, c in sorted(vals.items(), key=lambda kv: -med(kv[1]['GRBM_GUI_ACTIVE'])):
    w = med(c['SQ_WAVES']) or 1
    rd, wr = med(c['SQ_INSTS_VMEM_RD']), med(c['SQ_INSTS_VMEM_WR'])
    rq, wq = med(c['TCP_TCC_READ_REQ_sum']), med(c['TCP_TCC_WRITE_REQ_sum'])
    dur = med(c['GRBM_GUI_ACTIVE']) / 8.0
    print('%-44s %8d %9.1f %9.1f %10.0f %10.0f %8.1f %8.1f %10.0f %8.1f' % (k, w, rd / w, wr / w, rq, wq, rq / max(rd, 1), wq / max(wr, 1), med(c['TCP_TOTAL_CACHE_ACCESSES_sum']), 100 * med(c['TA_BUSY_avr']) / max(dur, 1)))
PY
