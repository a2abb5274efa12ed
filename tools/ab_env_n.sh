#!/bin/bash
# alternate environment settings on the headline step in one session: tools/ab_env_n.sh REPS "VAR=a" "VAR=b" ...
REPS=$1; shift
for rep in $(seq $REPS); do
  for v in "$@"; do
    echo -n "$v: "; env $v python bench.py --steps 192 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'p90', round(d['step_ms_spread']['p90'],4), 'scatter', round(d['roofline']['avg_launch_ms'],4))"
  done
done
