# GPU box, repo root: config 4 with the background's chains on their own stream beside the foreground's, against one stream; alternated
mkdir -p gpurun_out
O=gpurun_out/r6_ab_cfg4_bkg_stream.txt
: > $O
for rep in 1 2 3; do
  for v in 1 0; do
    ARCN_NEUS_BKG_STREAM=$v python bench.py --config neus_ngp_multivol --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('bkg_stream=$v rep=$rep ms_per_step %.4f device %s' % (r['ms_per_step'], r['config']['step_ms_device']))" >> $O
  done
done
cat $O
