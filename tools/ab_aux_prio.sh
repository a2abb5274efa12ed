#!/bin/bash
# priorities of the step's stream (ARCN_MAIN_PRIORITY) and of the sampling stream (ARCN_AUX_PRIORITY) of the headline step, alternating
python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())"
run() { echo -n "$*: "; env "$@" python bench.py --steps 192 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'gather in step', round(d['roofline_lookup']['avg_launch_ms'],4))"; }
for rep in 1 2; do
run ARCN_X=0
run ARCN_AUX_PRIORITY=1
run ARCN_AUX_PRIORITY=2
run ARCN_MAIN_PRIORITY=0 ARCN_AUX_PRIORITY=1
run ARCN_MAIN_PRIORITY=0
done
