mkdir -p gpurun_out
KEYS="march_count hashgrid_fwd" REPS=3 bash tools/ab_libs_keys.sh > gpurun_out/r5s4_ab_lattice.txt 2>&1
MATCH="march_count hashgrid_fwd" bash tools/ab_libs_prof.sh > gpurun_out/r5s4_prof_lattice.txt 2>&1
cat gpurun_out/r5s4_ab_lattice.txt gpurun_out/r5s4_prof_lattice.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ngp_reference.py tests/test_gpu_trajectory.py -q -m gpu -k "k3 or march or persistent or sampler or g21 or ngp or trajectory or loop" 2>&1 | tail -n 5
