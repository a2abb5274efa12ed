for v in 1 0 1 0; do
ARCN_NEUS_CORNERS=$v python bench.py --no-psnr --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('corners $v', 'headline', round(r['ms_per_step'],4), {k: round(v.get('ms_per_step',0),3) for k,v in r['other_configs'].items()})"
done
