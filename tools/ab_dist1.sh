#!/bin/bash
# the N > 1 code path of the headline step on ONE rank (ARCN_DIST_FORCE=1, RCCL communicator of size 1) next to the single-GPU step:
# what the two-pass optimiser form + the collectives' launch overhead cost before any byte goes over xGMI
run() { echo -n "$*: "; env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 192 --warmup 32 --no-cpu-baseline --no-other-configs --no-psnr 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), 'p50', round(d['step_ms_spread']['p50'],4), 'scatter', round(d['roofline']['avg_launch_ms'],4), d['rccl'] and {k: d['rccl'][k] for k in ('backend','collectives_per_step','allreduce_alone_ms')})"; }
for rep in 1 2; do
run ARCN_X=0
run ARCN_DIST_FORCE=1
run ARCN_DIST_FORCE=1 ARCN_GRAD_SEGMENTS=0
run ARCN_FUSE_ADAM=0
done
