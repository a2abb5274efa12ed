#!/bin/bash
# Run ON THE GPU BOX: per-kernel stats of the step on a one-rank RCCL communicator: flat all-reduce (default), level groups, sharded optimiser
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-dist1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for MODE in flat levels sharded; do
    env ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=$MODE timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/$MODE -o $MODE --output-format csv -- \
        python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29519 $ROOT/bench.py --gpus 1 --steps 64 --warmup 16 \
        --no-cpu-baseline --no-other-configs --no-psnr > $OUT/$MODE.json 2> $OUT/$MODE.log
    find $OUT/$MODE -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -25 {}' > $OUT/${MODE}_kernel_stats_top.csv
done
cd $ROOT
