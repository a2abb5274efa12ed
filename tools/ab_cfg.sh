#!/bin/bash
# alternate environment settings on one module-path bench config in one session: tools/ab_cfg.sh <config> REPS "VAR=a" "VAR=b" ...
CFG=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do
  for v in "$@"; do
    echo -n "$CFG $v: "; env $v python bench.py --config $CFG --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4))"
  done
done
