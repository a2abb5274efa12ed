for rep in 1 2 3; do
for w in 4096 3072 2560 2048 1536; do
ARCN_MARCH_WAVES=$w python bench.py --no-psnr --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('march_waves $w', 'step', round(r['ms_per_step'],4), 'p50', round(r['step_ms_spread']['p50'],4), 'lookup', round(r['roofline_lookup']['frac'],3), 'gather', round(r['kernel_ms']['hashgrid_fwd'],4), 'scatter', round(r['kernel_ms']['hashgrid_bwd'],4))"
done; done
