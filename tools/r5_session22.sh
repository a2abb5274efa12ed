python -m pytest tests/test_gpu_step_glue.py tests/test_gpu_composite.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -n 3
for rep in 1 2; do
python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg4 ms_per_step', round(d['ms_per_step'],4))"
done
bash tools/prof_config.sh r5e neus_ngp_multivol 2>&1 | grep -n "arcn share\|hashgrid" | cut -c1-190
