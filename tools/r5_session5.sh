mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -n 8
python bench.py > gpurun_out/r5s5_bench.json 2> gpurun_out/r5s5_bench.err; tail -c 600 gpurun_out/r5s5_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r5s5_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline'].get('frac'), r['roofline'].get('traffic_stale'))
for k, v in (r.get('other_configs') or {}).items():
    print(k, v.get('ms_per_step'), v.get('steps'), v.get('warmup'), v.get('error'))
print(r.get('psnr'))
PY
