# usage: tools/ab_env.sh VAR a b [reps]: alternate bench runs with VAR=a / VAR=b in one session (same box, same clocks)
VAR=$1; A=$2; B=$3; REPS=${4:-3}
for rep in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"
  done
done
