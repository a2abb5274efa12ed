"""Data-parallel training of the packed NGP pipeline, one process per GPU: what replaces the reference's
DistributedDataParallel + image-level DistributedSampler (common/trainer/basic_trainer.py:197-198, arcnerf/trainer/arcnerf_trainer.py:243-246)
for this path.  Per step: every rank takes its shard of the global ray batch - balanced by the per-ray sample counts of the batch's previous
visit (distributed.balanced_shards) once they are known, equal ray counts before - runs NgpPipeline.train_step with the level-grouped
gradient exchange chosen by ARCN_GRAD_SYNC (flat: one all-reduce, the default; levels: distributed.LevelGroupedGradSync overlapped with the
scatter; sharded: distributed.ShardedGradSync), and applies
the occupancy refresh every `epoch_optim` steps with rank 0's fields broadcast to all (distributed.broadcast_occupancy).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_ddp.py [steps=200] [rays=8192]
    (one process: python tools/train_ddp.py)

`train(...)` is importable: tests/test_gpu_distributed.py runs it on two gloo ranks sharing one GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arcnerf_amd import distributed as D  # noqa: E402
from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays  # noqa: E402


def make_pool(cfg, dev, n_rays, n_batches, seed=100):
    """global ray batches with an analytic target: rays that reach the `truth` occupancy see orange, the others the white background"""
    truth = torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, 0)).to(dev)
    probe = NgpPipeline(NgpField(cfg, device=dev, seed=1), max_rays=n_rays, max_samples=n_rays * 96)
    probe.set_bitfield(truth)
    pool = []
    for k in range(n_batches):
        o, d = synthetic_rays(n_rays, seed=seed + k, device=dev)
        probe.sample(o, d)
        hit = (probe.buf['counts'][:n_rays] > 1).float()[:, None]
        pool.append((o, d, (hit * torch.tensor([0.8, 0.3, 0.1], device=dev) + (1 - hit) * 1.0).contiguous(), torch.ones(n_rays, 3, device=dev)))
    return pool


def train(cfg, dev, rank, world, steps, n_rays, n_batches=4, sync='levels', balance=True, max_samples=None, level_cuts=(8,)):
    """-> dict(params, bitfield, opafield (cpu tensors), shard_log).  Every rank ends with the same parameters and occupancy."""
    fld = NgpField(cfg, device=dev, seed=0)
    D.broadcast_params(fld.params, src=0)
    pipe = NgpPipeline(fld, max_rays=n_rays, max_samples=max_samples or n_rays * 96, prefetch_depth=1)
    pipe.rng.set_state(D_rank_state(rank))
    pipe.occupancy_sync = D.broadcast_occupancy
    pool = make_pool(cfg, dev, n_rays, n_batches)
    counts_of = [None] * n_batches          # per-ray sample counts of the global batch at its previous visit (gathered from all ranks)
    grad_sync = all_reduce = None
    if world > 1 or D.forced():
        if sync == 'levels' and pipe.level_major:
            grad_sync = D.LevelGroupedGradSync(fld, level_cuts)
        elif sync == 'sharded':
            grad_sync = D.ShardedGradSync(fld.n_params, world, rank)
        else:
            all_reduce = lambda t: D.allreduce_grads(t, world)   # noqa: E731
    shard_log, loss = [], None
    for it in range(steps):
        k = it % n_batches
        o, d, tgt, bkg = pool[k]
        if balance and counts_of[k] is not None:
            b = D.balanced_shards(counts_of[k], world)
            lo, hi = b[rank], b[rank + 1]
        else:
            lo, hi = D.shard_range(n_rays, rank, world)
        hi = max(hi, lo + 1) if lo < n_rays else hi
        shard_log.append((lo, hi))
        # unequal shards: the rank's mean over its rays, weighted by its share of the global batch (the optimiser divides the SUM by `world`)
        loss = pipe.train_step(o[lo:hi].contiguous(), d[lo:hi].contiguous(), tgt[lo:hi].contiguous(), bkg_color=bkg[lo:hi].contiguous(),
                               all_reduce=all_reduce, world_size=world, grad_sync=grad_sync, loss_scale=(hi - lo) * world / float(n_rays))
        if balance and world > 1:
            # this step's per-ray counts of the GLOBAL batch: every rank contributes its shard's (rays it did not march: 0)
            full = torch.zeros(n_rays, dtype=torch.int32, device=dev)
            pipe.aux_stream.synchronize()
            full[lo:hi] = pipe.buf['counts'][:hi - lo]
            torch.distributed.all_reduce(full)
            counts_of[k] = full
        pipe.update_occupancy(it + 1, apply=True)
    torch.cuda.synchronize()
    return {'params': fld.params.cpu(), 'bitfield': pipe.bitfield.cpu(), 'opafield': pipe.opafield.cpu(), 'shards': shard_log,
            'loss': float(loss) if loss is not None else None, 'steps': pipe.step_count}


def D_rank_state(rank):
    """rank-local sampler stream (the reference's generators are rank-local too)"""
    from arcnerf_amd.ops import functional as F
    return F.Pcg32Host(9121 + rank).state


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    rank, world, local = D.env_world()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    D.init_from_env(device=dev)
    out = train(NgpConfig(white_bkg=True), dev, rank, world, steps, n_rays, sync=os.environ.get('ARCN_GRAD_SYNC', 'flat'))
    digest = float(out['params'].double().abs().sum())
    agree = D.max_over_ranks(digest, device=dev) == digest
    if rank == 0:
        print(json.dumps({'world': world, 'steps': out['steps'], 'loss': out['loss'], 'param_abs_sum': digest, 'occupied': float(out['bitfield'].float().mean()),
                          'last_shards_rank0': out['shards'][-1], 'replicas_agree': bool(agree)}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
