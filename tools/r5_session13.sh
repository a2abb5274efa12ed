mkdir -p gpurun_out
python -m pytest tests/test_gpu_composite.py -q -m gpu -x -k fused_neus 2>&1 | tail -n 12
for rep in 1 2 3; do for v in 0 1; do
ARCN_FUSE_ADAM=$v python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fuse_adam=$v cfg4 ms_per_step', round(d['ms_per_step'],4))"
done; done
