# the round's closing sequence on the GPU box: the GPU suite, the default bench line, then tools/profile_round.sh (fresh counters for these
# sources -> profiles/pmc_traffic.json), the default line once more on those counters, the config-4 profile
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -n 3
bash tools/profile_round.sh r6z 2>&1 | tail -n 5
cp gpurun_out/r6z/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > gpurun_out/r6z_bench_default.json 2> gpurun_out/r6z_bench_default.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r6z_bench_default.json').read().strip().splitlines()[-1])
print('headline', r['value'], r['ms_per_step'], 'roofline', r['roofline']['frac'], 'stale', r['roofline'].get('traffic_stale'), 'traffic', r['roofline'].get('traffic'), 'lookup', r['roofline_lookup']['frac'])
print({k: round(v.get('ms_per_step', 0), 4) for k, v in r['other_configs'].items()})
print(r['psnr_analytic_scene']['psnr_at_iter'], r['psnr_analytic_scene']['training_loop'], r['cpu_baseline']['value'])
print(r['step_ms_spread'])
PY
bash tools/prof_config.sh r6z neus_ngp_multivol 2>&1 | tail -n 32
