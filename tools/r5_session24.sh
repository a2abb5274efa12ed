for rep in 1 2; do for v in 0 1; do
ARCN_NEUS_CORNERS=$v ARCN_OTHER_CONFIGS=ngp_module,neus_ngp_multivol python bench.py --no-psnr --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('corners $v', 'headline', round(r['ms_per_step'],4), 'stale', r['roofline']['traffic_stale'], {k: round(v.get('ms_per_step',0),3) for k,v in r['other_configs'].items()})"
ARCN_NEUS_CORNERS=$v python bench.py --config neus_ngp_multivol --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  standalone corners=$v', round(d['ms_per_step'],4))"
done; done
