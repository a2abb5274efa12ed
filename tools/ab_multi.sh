# usage: tools/ab_multi.sh REPS "ENV1=a ENV2=b" "ENV1=c" ...   (each quoted argument is one configuration; '-' = defaults)
REPS=$1; shift
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$cfg]', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
