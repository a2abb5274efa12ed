for order in "ngp_module,nerf,neus,neus_ngp_multivol,hdrnerf" "ngp_module,neus_ngp_multivol,nerf,neus,hdrnerf" "neus_ngp_multivol"; do
ARCN_OTHER_CONFIGS=$order python bench.py --no-psnr --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.readline()); print('$order', {k: round(v.get('ms_per_step',0),3) for k,v in r['other_configs'].items()})"
done
python bench.py --config neus_ngp_multivol --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('standalone', round(d['ms_per_step'],4))"
