# config 4 A/B: the module path (sampler prefetch on) / trainer.FusedNeusNgpStep, alternating; then the profile of the default
for rep in 1 2; do
  for v in eager fused; do
    ARCN_MODULE_STEP=$v python bench.py --config neus_ngp_multivol --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$v ms_per_step', round(d['ms_per_step'],4), 'fg pts', d['config']['samples_per_step_per_gpu'], 'bkg', d['roofline']['bkg_samples_per_step'])"
  done
done
bash tools/prof_config.sh r5b neus_ngp_multivol 2>&1 | tail -42
