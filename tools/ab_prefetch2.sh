run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; print('$*', round(d['ms_per_step'],4), 'fwd', round(k['hashgrid_fwd'],4), 'bwd', round(k['hashgrid_bwd'],4))"; }
for rep in 1 2; do
for at in 0 1 2 3 4; do run ARCN_PREFETCH_DEPTH=2 ARCN_PREFETCH_AT=$at; done
run ARCN_PREFETCH_DEPTH=1 ARCN_PREFETCH_AT=1
run ARCN_PREFETCH_DEPTH=1 ARCN_PREFETCH_AT=2
done
