mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_glue.py tests/test_gpu_composite.py -q -m gpu -x 2>&1 | tail -n 25
python tools/exp_cfg4_fused_ops.py > gpurun_out/r5_cfg4_fused_ops2.txt 2>&1; grep -v "^\[W\|Warning\|warn" gpurun_out/r5_cfg4_fused_ops2.txt | tail -n 90
