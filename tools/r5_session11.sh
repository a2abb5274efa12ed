mkdir -p gpurun_out
for rep in 1 2 3; do
for v in 0 1; do
ARCN_SDF_NOGRAD_FAST=$v python bench.py --config neus --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nograd_fast=$v neus ms_per_step', round(d['ms_per_step'],3))"
done
done
python tools/exp_neus_ops.py > gpurun_out/r5_neus_ops2.txt 2>&1; grep -n "device kernels\|aten ops with device" -A 14 gpurun_out/r5_neus_ops2.txt | cut -c1-170 | head -60; tail -n 1 gpurun_out/r5_neus_ops2.txt
