#!/bin/bash
# GPU box: kernel timeline of one steady step of a bench config: tools/trace_step.sh <config> [marker kernel substring]
CFG=${1:-ngp_module}; MARK=${2:-scatter_accum}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts
timeout 300 rocprofv3 --kernel-trace -d /tmp/ts -o ts --output-format csv -- python $ROOT/bench.py --config $CFG --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --no-psnr > /dev/null 2>&1
python - "$MARK" <<'PY'
import csv, re, glob, sys
f=glob.glob('/tmp/ts/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def short(n):
    n=re.sub(r'^void ','',n); n=n.replace('arcn::','').replace('at::native::','')
    return re.sub(r'\(.*','',n)[:70]
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),r['Queue_Id']) for r in rows)
idx=[i for i,e in enumerate(ev) if sys.argv[1] in e[2]]
a=idx[-5]; b=idx[-4]
t0=ev[a][1]; prev=t0
for e in ev[a+1:b+1]:
    print('%9.1f %9.1f  dur %7.1f gap %6.1f q%s %s'%((e[0]-t0)/1e3,(e[1]-t0)/1e3,(e[1]-e[0])/1e3,(e[0]-prev)/1e3,e[3],e[2]))
    prev=max(prev,e[1])
PY
