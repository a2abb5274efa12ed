#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  tools/profile_round.sh <tag>
# Three separate rocprofv3 passes of the same bench command: kernel trace + stats, PMC FETCH_SIZE, PMC WRITE_SIZE
# (counters are never combined with other trace domains).  Everything lands in gpurun_out/<tag>/.
set -u
TAG=${1:-r1x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- \
    python $ROOT/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C -d $OUT/pmc_$C -o $TAG --output-format csv -- \
        python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc_$C.log
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_MFMA -o $TAG --output-format csv -- \
    python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/pmc_MFMA.log
cd $ROOT
python tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json
ls -R $OUT | head -40
