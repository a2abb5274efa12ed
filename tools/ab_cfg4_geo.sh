# GPU box, repo root: config 4 with both geometry nets fused (arcn_geo2_*) against round 5's chains, alternated; -> gpurun_out/r6_ab_cfg4_geo.txt
mkdir -p gpurun_out
O=gpurun_out/r6_ab_cfg4_geo.txt
: > $O
for rep in 1 2 3; do
  for v in 1 0; do
    ARCN_NEUS_FUSED_GEO=$v python bench.py --config neus_ngp_multivol --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('fused_geo=$v rep=$rep ms_per_step %.4f device_p50 %.4f samples/s %.4g' % (r['ms_per_step'], r['config']['step_ms_device']['p50'], r['value']))" >> $O
  done
done
cat $O
