#!/bin/bash
# ms per step of the wide-net configurations (BASELINE configs 1 / 3 / 5), one line each: tools/bench_wide.sh [configs...]
for m in ${@:-nerf neus hdrnerf}; do
  python bench.py --config $m --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['ms_per_step'], 3), 'ms/step')"
done
