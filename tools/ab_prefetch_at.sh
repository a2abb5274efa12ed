# GPU box, repo root: the headline with the coming batch's marching queued at the step's points 1 .. 5 (NgpPipeline._prefetch_point): step time and the
# gather's in-step roofline fraction
mkdir -p gpurun_out
O=gpurun_out/r6_ab_prefetch_at.txt
: > $O
for rep in 1 2; do
  for v in 4 3 2 1 5; do
    ARCN_PREFETCH_AT=$v python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.readline()); k=r.get('kernel_ms') or {}; print('prefetch_at=$v rep=$rep ms_per_step %.4f lookup_in_step %.3f scatter %.3f' % (r['ms_per_step'], r['roofline_lookup']['frac'], r['roofline']['frac']), {a: round(b, 4) for a, b in k.items()} if isinstance(k, dict) else '')" >> $O
  done
done
cat $O
