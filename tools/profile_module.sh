#!/bin/bash
# Run ON THE GPU BOX from the repo root: tools/profile_module.sh <tag> <config>: bench JSONs of the module-path configs and a
# rocprofv3 kernel-trace + stats pass of one of them (counters of the dominant product: tools/pmc_kernel.sh, separate passes).
set -u
TAG=${1:-r2f}; CFG=${2:-nerf}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for c in nerf hdrnerf neus neus_ngp_multivol; do
  extra="--no-cpu-baseline"; [ "$c" == "nerf" ] && extra=""
  python bench.py --config $c --steps 8 --warmup 3 $extra 2>/dev/null | tail -1 > $OUT/bench_$c.json
done
python bench.py --config nerf --steps 8 --warmup 3 --no-cpu-baseline --chunk-pts 1048576 2>/dev/null | tail -1 > $OUT/bench_nerf_chunk1m.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
    python $ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_${CFG}_under_rocprof.json 2> $OUT/trace.log
cd $ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${CFG}_kernel_stats.csv
bash tools/pmc_kernel.sh "tools/exp_gemm_one.py nt" gemm_rows_split "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE" > $OUT/pmc_gemm_nt_split.txt 2>&1
bash tools/pmc_kernel.sh "tools/exp_gemm_one.py tn" gemm_tn_split "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE" > $OUT/pmc_gemm_tn_split.txt 2>&1
bash tools/exp_clock.sh nt 1 > $OUT/clock_nt_split.txt 2>&1
bash tools/exp_clock.sh nt 0 > $OUT/clock_nt_exact.txt 2>&1
rm -rf $OUT/trace
ls $OUT
