# the round's closing sequence on the GPU box: the GPU suite, the default bench line, then tools/profile_round.sh (LAST: fresh counters)
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -n 3
python bench.py > gpurun_out/r6z_bench_default.json 2> gpurun_out/r6z_bench_default.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r6z_bench_default.json').read().strip().splitlines()[-1])
print('headline', r['value'], r['ms_per_step'], 'roofline', r['roofline']['frac'], 'stale', r['roofline'].get('traffic_stale'), 'lookup', r['roofline_lookup']['frac'])
print({k: round(v.get('ms_per_step', 0), 4) for k, v in r['other_configs'].items()})
print(r['psnr_analytic_scene']['psnr_at_iter'], r['cpu_baseline']['value'])
PY
bash tools/profile_round.sh r6z 2>&1 | tail -n 5
python bench.py --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('after profile: stale', r['roofline'].get('traffic_stale'), 'traffic', r['roofline'].get('traffic'), 'frac', r['roofline']['frac'])"
bash tools/prof_config.sh r6z neus_ngp_multivol 2>&1 | tail -n 3
