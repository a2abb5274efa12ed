mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -n 4
bash tools/prof_config.sh r5c neus 2>&1 | head -3
