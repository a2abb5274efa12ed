run() { env "$@" python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['config']['samples_per_step_per_gpu'])"; }
for rep in 1 2 3; do
run ARCN_PREFETCH_DEPTH=1 ARCN_PREFETCH_AT=1
run ARCN_PREFETCH_DEPTH=2 ARCN_PREFETCH_AT=3
run ARCN_PREFETCH_DEPTH=2 ARCN_PREFETCH_AT=4
done
