for q in 4 8 4 8; do
GPU_MAX_HW_QUEUES=$q ARCN_OTHER_CONFIGS=ngp_module,neus_ngp_multivol python bench.py --no-psnr --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('queues $q', 'headline', round(r['ms_per_step'],4), {k: round(v.get('ms_per_step',0),3) for k,v in r['other_configs'].items()})"
done
