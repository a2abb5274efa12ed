# like ab_multi.sh but with bench.py's default step counts (what the driver runs)
REPS=$1; shift
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    env $e python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$cfg]', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline_lookup']['avg_launch_ms'],4))"
  done
done
