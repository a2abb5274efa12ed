# GPU box, repo root: the headline with both nets' forward in one kernel (arcn_ngp_nets_fwd) against the two launches, alternated
mkdir -p gpurun_out
O=gpurun_out/r6_ab_fused_nets.txt
: > $O
for rep in 1 2 3; do
  for v in 1 0; do
    ARCN_FUSED_NETS=$v python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('fused_nets=$v rep=$rep ms_per_step %.4f samples/s %.4g roofline %.3f lookup %.3f' % (r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline_lookup']['frac']))" >> $O
  done
done
cat $O
