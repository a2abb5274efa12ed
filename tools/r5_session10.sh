mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_glue.py -q -m gpu -x -k "row or concat2 or graph_free" 2>&1 | tail -n 12
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_fullwidth.py -q -m gpu -x -k "sdf_chain or neus or fullwidth or radiance_chain" 2>&1 | tail -n 6
python bench.py --config neus --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('neus ms_per_step', d['ms_per_step'])"
