#!/bin/bash
# Counters of the hash gather INSIDE the training step (bench.py), with the marcher of a later batch next to it (default schedule) and
# without (ARCN_PREFETCH_AT=0: the marching is issued right after the forward), and alone (tools/exp_gather.py) - counters only, one
# --pmc set per pass.  usage (GPU box, repo root): tools/pmc_gather_instep.sh <out-file>
OUT=$1
ROOT=$(pwd)
SETS=("TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES")
cd /tmp && export TMPDIR=/tmp
: > $OUT
for MODE in "ARCN_X=0" "ARCN_PREFETCH_AT=0" "ALONE=1"; do
  i=0
  rm -rf /tmp/pgi_*
  for SET in "${SETS[@]}"; do
    i=$((i+1))
    if [ "$MODE" = "ALONE=1" ]; then
      timeout 300 rocprofv3 --pmc $SET -d /tmp/pgi_$i -o pk --output-format csv -- python $ROOT/tools/exp_gather.py > /dev/null 2>&1
    else
      env $MODE timeout 300 rocprofv3 --pmc $SET -d /tmp/pgi_$i -o pk --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs --no-psnr > /dev/null 2>&1
    fi
  done
  echo "== $MODE" >> $OUT
  python - >> $OUT <<'PY'
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob('/tmp/pgi_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hashgrid_fwd_bal_kernel' in r['Kernel_Name']:
            vals[int(r.get('Grid_Size', 0) or 0)][r['Counter_Name']].append(float(r['Counter_Value']))
grid = max(vals, key=lambda g: sum(len(v) for v in vals[g].values()))      # the training-sized launches (the refresh launches another size)
c = {k: sorted(v)[len(v) // 2] for k, v in vals[grid].items()}
for k in sorted(c):
    print('    %-34s median %14.1f  (n=%d)' % (k, c[k], len(vals[grid][k])))
if 'TCC_HIT_sum' in c:
    print('    L2 hit rate %.3f; requests per launch %.3g; latency per request %.0f cycles; TCP pending-stall share %.2f; wait-inst share %.2f' % (
        c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), c.get('TCP_TCC_READ_REQ_sum', 0), c.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / max(1.0, c.get('TCP_TCC_READ_REQ_sum', 1)),
        c.get('TCP_PENDING_STALL_CYCLES_sum', 0) / max(1.0, c.get('TCP_GATE_EN1_sum', 1)), c.get('SQ_WAIT_INST_ANY', 0) / max(1.0, c.get('SQ_WAVE_CYCLES', 1))))
PY
done
