#!/bin/bash
# A/B of an environment setting on the whole bench step in one session: tools/ab_env_bench.sh "VAR=a" "VAR=b" [reps]
A=$1; B=$2; REPS=${3:-3}
for rep in $(seq $REPS); do
  for v in "$A" "$B"; do
    echo -n "$v: "; env $v python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print(round(d['ms_per_step'],4), 'fwd', round(k['hashgrid_fwd'],4), 'bwd', round(k['hashgrid_bwd'],4))"
  done
done
