VARIANTS="base march8" KEYS="hashgrid_fwd hashgrid_bwd" REPS=3 PROF=1 MATCH="march_count" bash tools/ab_libs_multi.sh 2>&1 | tail -n 12
cd arcnerf_amd/lib; cp libarcnerf_hip.so keep.so; cd ../..
for rep in 1 2 3; do for v in base dir8; do cp arcnerf_amd/lib/alt_$v.so arcnerf_amd/lib/libarcnerf_hip.so
python bench.py --config neus_ngp_multivol --steps 48 --warmup 12 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v cfg4 ms_per_step', round(d['ms_per_step'],4))"
done; done
cp arcnerf_amd/lib/keep.so arcnerf_amd/lib/libarcnerf_hip.so; rm arcnerf_amd/lib/keep.so
