#!/bin/bash
# the N > 1 code path of the headline step on ONE rank (ARCN_DIST_FORCE=1, RCCL communicator of size 1) next to the single-GPU step:
# what the two-pass optimiser form + the collectives' launch overhead cost before any byte goes over xGMI.
#   ARCN_DIST_FORCE=1                                         the default exchange: ONE flat all-reduce after the backward (north_star's form)
#   ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=levels                   two level groups overlapped with the scatter, optimiser per group
#   ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=levels ARCN_GRAD_LEVEL_CUTS=11,5   three groups
#   ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=sharded                  reduce-scatter + optimiser on 1/N + all-gather
#   ARCN_FUSE_ADAM=0                            no collective at all, two-pass optimiser form
run() { echo -n "$*: "; env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 192 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['rccl']; print(round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'scatter', round(d['roofline']['avg_launch_ms'],4), r and {k: r.get(k) for k in ('backend','collectives_per_step','allreduce_alone_ms','exposed_ms','grad_sync')})"; }
for rep in 1 2; do
run ARCN_X=0
run ARCN_DIST_FORCE=1
run ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=levels
run ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=levels ARCN_GRAD_LEVEL_CUTS=11,5
run ARCN_DIST_FORCE=1 ARCN_GRAD_SYNC=sharded
run ARCN_FUSE_ADAM=0
done
