for cfg in neus nerf hdrnerf; do
for c in 0 262144 1048576; do
python bench.py --config $cfg --steps 24 --warmup 6 --no-cpu-baseline --chunk-pts $c 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$cfg chunk $c ms', round(r['ms_per_step'],3), r['config'].get('chunk_pts'))"
done
done
