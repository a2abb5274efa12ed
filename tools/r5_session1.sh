# round 5, first GPU session: the new tests, then the marcher A/B (closed-form lattice vs the systolic chain)
mkdir -p gpurun_out
ARCN_TRAJ_REPORT=1 python -m pytest tests/test_gpu_psnr.py -q -m gpu -s 2>&1 | tail -80 > gpurun_out/r5s1_psnr.txt
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_trajectory.py tests/test_gpu_distributed.py tests/test_gpu_ngp_reference.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5s1_tests.txt
KEYS="march_count hashgrid_fwd" REPS=3 bash tools/ab_libs_keys.sh > gpurun_out/r5s1_ab_lattice.txt 2>&1
MATCH="march_count hashgrid_fwd" bash tools/ab_libs_prof.sh > gpurun_out/r5s1_prof_lattice.txt 2>&1
tail -5 gpurun_out/r5s1_psnr.txt gpurun_out/r5s1_tests.txt; cat gpurun_out/r5s1_ab_lattice.txt gpurun_out/r5s1_prof_lattice.txt
