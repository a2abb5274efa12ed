#!/bin/bash
# ARCN_MARCH_WAVES=P: the marcher of the batch two steps ahead as P persistent wavefronts (wave w marches rays w, w + P, ...; 0: one wavefront
# per ray), alternating in one session.   usage: tools/ab_march_persist.sh [P ...]
run() { echo -n "$*: "; env "$@" python bench.py --steps 192 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'gather in step', round(d['roofline_lookup']['avg_launch_ms'],4), 'scatter', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"; }
PS=${@:-"0 4096"}
for rep in 1 2 3 4; do
for p in $PS; do run ARCN_MARCH_WAVES=$p; done
done
