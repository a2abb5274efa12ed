python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_composite.py tests/test_gpu_trajectory.py tests/test_gpu_switches.py -q -m gpu -x 2>&1 | tail -n 3
for rep in 1 2; do
python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg4 ms_per_step', round(d['ms_per_step'],4))"
done
