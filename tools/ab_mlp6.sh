cp arcnerf_amd/lib/libarcnerf_hip.so arcnerf_amd/lib/keep.so
cp arcnerf_amd/lib/alt_new.so arcnerf_amd/lib/libarcnerf_hip.so
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests/test_gpu_ngp_reference.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -3
cp arcnerf_amd/lib/keep.so arcnerf_amd/lib/libarcnerf_hip.so; rm arcnerf_amd/lib/keep.so
bash tools/ab_script.sh "bench.py --no-cpu-baseline --no-other-configs --steps 96 --warmup 16" 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        tag, js = l.split(': ', 1); d = json.loads(js); print(tag, d['value'], d['ms_per_step'])
    except Exception as e: print(l[:200])"
