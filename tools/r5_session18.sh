python -m pytest tests/test_gpu_kernels.py tests/test_gpu_composite.py tests/test_gpu_models.py tests/test_gpu_switches.py -q -m gpu -x 2>&1 | tail -n 4
for rep in 1 2; do
python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg4 ms_per_step', round(d['ms_per_step'],4))"
done
bash tools/prof_config.sh r5d neus_ngp_multivol 2>&1 | grep -n "arcn share\|scatter\|ms_per_step" | cut -c1-200
