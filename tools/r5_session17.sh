for rep in 1 2; do
for c in 0 64 128 192; do
ARCN_AUX_CUS=$c python bench.py --no-psnr --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('aux_cus $c', 'step', round(r['ms_per_step'],4), 'lookup', round(r['roofline_lookup']['frac'],3), 'gather_ms', round(r['kernel_ms']['hashgrid_fwd'],4), 'scatter', round(r['kernel_ms']['hashgrid_bwd'],4), 'p50', round(r['step_ms_spread']['p50'],4))"
done; done
