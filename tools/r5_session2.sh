mkdir -p gpurun_out
ARCN_TRAJ_REPORT=1 python -m pytest tests/test_gpu_psnr.py -q -m gpu -s -k "trains_like" > gpurun_out/r5s2_psnr_full.txt 2>&1
grep -n "followed\|reference'\|all-white\|^E  \|passed\|failed" gpurun_out/r5s2_psnr_full.txt | cut -c1-400 | head -60
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_trajectory.py tests/test_gpu_distributed.py tests/test_gpu_ngp_reference.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r5s2_tests.txt
tail -n 15 gpurun_out/r5s2_tests.txt
python bench.py --no-cpu-baseline --no-other-configs --no-psnr > gpurun_out/r5s2_bench.txt 2>&1; tail -n 20 gpurun_out/r5s2_bench.txt | cut -c1-600
