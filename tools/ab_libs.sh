cd arcnerf_amd/lib
for rep in 1 2 3; do
  for v in old new; do
    cp alt_$v.so libarcnerf_hip.so
    (cd ../..; python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))")
  done
done
cp alt_new.so libarcnerf_hip.so
