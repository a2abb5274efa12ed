for rep in 1 2 3 4; do for v in 0 1; do
ARCN_NEUS_CORNERS=$v python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('corners=$v cfg4 ms_per_step', round(d['ms_per_step'],4))"
done; done
