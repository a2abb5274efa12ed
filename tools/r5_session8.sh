mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_glue.py tests/test_gpu_composite.py -q -m gpu -x 2>&1 | tail -n 4
bash tools/prof_config.sh r5b neus_ngp_multivol 2>&1 | grep -n "arcn share\|blend\|jac_dz2\|sum_scale\|geo_out\|step_prep\|ms_per_step" | cut -c1-260
