ARCN_BENCH_TRACE=1 python bench.py --config neus_ngp_multivol --steps 12 --warmup 8 --no-cpu-baseline 2>&1 >/dev/null | grep TRACE | cut -c1-700
bash tools/prof_config.sh r5a neus_ngp_multivol 2>&1 | tail -45
