mkdir -p gpurun_out
ARCN_TRAJ_REPORT=1 python -m pytest tests/test_gpu_psnr.py -q -m gpu -s > gpurun_out/r5s3_psnr_full.txt 2>&1
grep -n "followed\|reference'\|all-white\|^E  \|passed\|failed" gpurun_out/r5s3_psnr_full.txt | cut -c1-420 | head -60
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -n 15
