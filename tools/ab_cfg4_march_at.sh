# GPU box, repo root: config 4, where the coming batches' samplers are queued inside the step; -> gpurun_out/r6_ab_cfg4_march_at.txt
mkdir -p gpurun_out
O=gpurun_out/r6_ab_cfg4_march_at.txt
: > $O
for rep in 1 2; do
  for v in blend start fg_bwd opt; do
    ARCN_NEUS_MARCH_AT=$v python bench.py --config neus_ngp_multivol --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('march_at=$v rep=$rep ms_per_step %.4f device_p50 %.4f' % (r['ms_per_step'], r['config']['step_ms_device']['p50']))" >> $O
  done
done
ARCN_PREFETCH_SAMPLES=0 python bench.py --config neus_ngp_multivol --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('no prefetch ms_per_step %.4f device_p50 %.4f' % (r['ms_per_step'], r['config']['step_ms_device']['p50']))" >> $O
cat $O
