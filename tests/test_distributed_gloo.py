"""world_size-2 `gloo` tests (CPU) of the data-parallel layer: ray sharding covers every ray exactly once, the single
flat-gradient all-reduce reproduces the single-process full-batch gradient, the bitfield/parameter broadcasts make replicas
identical, and replicas stay bit-identical after the optimiser step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arcnerf_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_grad(rays, params):
    """a deterministic stand-in for 'gradient of the mean loss over these rays' (linear in per-ray contributions)"""
    w = torch.sin(rays.sum(-1, keepdim=True) * torch.arange(1, params.numel() + 1)[None] * 0.01)
    return (w * (1.0 + params[None])).sum(0)


def _worker(rank, world, port, n_rays, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rays = torch.rand(n_rays, 6, generator=g)  # every rank sees the same global ray list and takes its shard
    params = torch.linspace(-1, 1, 1000) if rank == 0 else torch.zeros(1000)
    D.broadcast_params(params, src=0)
    lo, hi = D.shard_range(n_rays, rank, world)
    grads = _fake_grad(rays[lo:hi], params)
    D.allreduce_grads(grads, world)
    # the reference folds 1/world into DDP's averaging; here it is the optimiser's grad_scale
    params = params - 0.1 * grads * (1.0 / n_rays)
    bits = (torch.arange(4096) % 7 == 0).to(torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
    D.broadcast_bitfield(bits, src=0)
    t = D.max_over_ranks(1.0 + rank)
    ret[rank] = (lo, hi, grads.numpy(), params.numpy(), bits.numpy(), t)
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_single_process():
    world, n_rays = 2, 1001
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (lo0, hi0, g0, p0, b0, t0), (lo1, hi1, g1, p1, b1, t1) = ret[0], ret[1]
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)
    g = torch.Generator().manual_seed(0)
    rays = torch.rand(n_rays, 6, generator=g)
    ref = _fake_grad(rays, torch.linspace(-1, 1, 1000)).numpy()
    np.testing.assert_allclose(g0, ref, rtol=1e-5, atol=1e-4)
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)  # replicas stay bit-identical
    assert np.array_equal(b0, b1) and b0.sum() == len(range(0, 4096, 7))
    assert t0 == t1 == 2.0


@pytest.mark.parametrize('n,world', [(10, 3), (7, 8), (32768, 8), (0, 2), (5, 1)])
def test_shard_range_partitions_exactly(n, world):
    seen = []
    for r in range(world):
        lo, hi = D.shard_range(n, r, world)
        assert 0 <= lo <= hi <= n
        seen += list(range(lo, hi))
    assert seen == list(range(n))
    sizes = [D.shard_range(n, r, world)[1] - D.shard_range(n, r, world)[0] for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def test_balanced_shards_equalise_sample_counts():
    g = torch.Generator().manual_seed(1)
    counts = (torch.rand(8192, generator=g) ** 4 * 300).long()  # a few long rays, many short ones
    counts[::5] = 0
    b = D.balanced_shards(counts, 8)
    assert b[0] == 0 and b[-1] == 8192 and all(b[i] <= b[i + 1] for i in range(8))
    per = [int(counts[b[i]:b[i + 1]].sum()) for i in range(8)]
    assert max(per) - min(per) <= int(counts.max()) + 1
    naive = [int(counts[D.shard_range(8192, r, 8)[0]:D.shard_range(8192, r, 8)[1]].sum()) for r in range(8)]
    assert max(per) - min(per) <= max(naive) - min(naive)
    assert D.balanced_shards(torch.zeros(0), 4) == [0, 0, 0, 0, 0]


def test_single_process_helpers_are_noops():
    t = torch.ones(4)
    assert D.allreduce_grads(t, 1) is t and D.broadcast_bitfield(t) is t and D.broadcast_params(t) is t
    assert D.max_over_ranks(3.5) == 3.5


def _adam(p, g, m, v, step, scale):
    """elementwise Adam on (views of) flat buffers, clearing the gradient it reads - what the fused optimiser kernel does per element"""
    g = g * scale
    m.mul_(0.9).add_(g, alpha=0.1)
    v.mul_(0.999).addcmul_(g, g, value=0.001)
    p.sub_(0.01 * (m / (1 - 0.9 ** step)) / ((v / (1 - 0.999 ** step)).sqrt() + 1e-8))


def _sharded_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    D.init_from_env(backend='gloo')
    n = 100003
    sync = D.ShardedGradSync(n)
    assert (sync.world, sync.rank) == (world, rank) and sync.per % 4 == 0 and 0 <= n - sync.body < 4 * world
    runs = {}
    for form in ('flat', 'sharded'):
        p = torch.linspace(-1, 1, n)
        m, v, grads = torch.zeros(n), torch.zeros(n), torch.zeros(n)
        for step in range(1, 41):
            g = torch.Generator().manual_seed(1000 * rank + step)
            grads += torch.randn(n, generator=g) * (1.0 + p.abs())          # this rank's contribution depends on the parameters: a divergence would grow
            if form == 'flat':
                D.allreduce_grads(grads, world)
                _adam(p, grads, m, v, step, 1.0 / world)
                grads.zero_()
            else:       # what NgpPipeline.train_step(grad_sync=ShardedGradSync) does
                sync.launch(grads)
                sync.wait()
                for lo, hi in sync.segments:
                    _adam(p[lo:hi], grads[lo:hi], m[lo:hi], v[lo:hi], step, 1.0 / world)
                    grads[lo:hi].zero_()
                sync.clear_foreign(grads)
                sync.gather(p)
                assert float(grads.abs().max()) == 0.0
        foreign = float(m[:sync.lo].abs().sum() + m[sync.hi:sync.body].abs().sum())
        if form == 'sharded':
            # a checkpoint needs every shard's moments: not current after sharded steps, complete after the collective gather
            was_current = sync.moments_current
            sync.gather_moments(m, v)
            runs['moments'] = (was_current, sync.moments_current, m.numpy().copy(), v.numpy().copy())
        else:
            runs['flat_moments'] = (m.numpy().copy(), v.numpy().copy())
        runs[form] = (p.numpy().copy(), foreign)
    ret[rank] = (runs, sync.segments, sync.body)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_gradient_sync_equals_flat_allreduce(world):
    """distributed.ShardedGradSync (reduce-scatter, the optimiser on this rank's shard, all-gather): after 40 steps every rank holds the
    SAME parameters; on two ranks they are the flat all-reduce form's bit for bit (a + b in either order), on four the two forms sum an
    element's four contributions in different orders (the ring's chunk boundaries move with the buffer length) and agree to the rounding of
    those sums; the shards tile the buffer once, and a rank never touches the moments of another rank's shard"""
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    ref = ret[0][0]['flat'][0]
    spans = []
    for r in range(world):
        runs, segments, body = ret[r]
        assert np.array_equal(runs['flat'][0], ref) and np.array_equal(runs['sharded'][0], ret[0][0]['sharded'][0]), r
        if world == 2:
            assert np.array_equal(runs['sharded'][0], ref), r
        else:
            assert np.abs(runs['sharded'][0] - ref).max() <= 1e-4 and np.mean(runs['sharded'][0] != ref) < 0.5
        assert runs['sharded'][1] == 0.0 and runs['flat'][1] > 0.0
        was_current, now_current, m_all, v_all = runs['moments']
        assert not was_current and now_current
        assert np.array_equal(m_all, ret[0][0]['moments'][2]) and np.array_equal(v_all, ret[0][0]['moments'][3]), r     # the same complete state on every rank
        fm, fv = runs['flat_moments']
        if world == 2:
            assert np.array_equal(m_all, fm) and np.array_equal(v_all, fv)      # ... and it is the replicated optimiser's
        else:
            assert np.abs(m_all - fm).max() <= 1e-4 * max(1.0, np.abs(fm).max()) and np.abs(v_all - fv).max() <= 1e-4 * max(1.0, np.abs(fv).max())
        spans.append(segments[0])
        assert segments[1:] == ([(body, 100003)] if body < 100003 else [])
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == ret[0][2] and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _level_sync_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    D.init_from_env(backend='gloo')
    from arcnerf_amd.pipeline import NgpConfig, NgpField
    fld = NgpField(NgpConfig(hashmap_size=12), device='cpu', seed=0)
    n = fld.n_params
    g = torch.Generator().manual_seed(300 + rank)
    grads = torch.randn(n, generator=g)
    flat = grads.clone()
    D.allreduce_grads(flat, world)
    out = {}
    for cuts in ((8,), (11, 5), (3,)):
        sync = D.LevelGroupedGradSync(fld, cuts)
        seg = grads.clone()
        for gi in range(len(sync.groups)):           # what NgpPipeline.backward does: scatter group gi, then put its slice on the wire
            sync.launch_group(gi, seg)
        for gi in range(len(sync.groups)):
            sync.wait(gi)
        out[cuts] = (seg.numpy(), sync.groups, sync.segments)
    ret[rank] = (flat.numpy(), out, n, list(fld.offsets))
    dist.destroy_process_group()


def test_level_grouped_gradient_sync_equals_flat_allreduce():
    """distributed.LevelGroupedGradSync (the N > 1 step's default: the table gradient leaves in level groups while the scatter is still
    running): the groups' level masks cover every level once, their slices tile the flat buffer exactly once (the MLP weights ride with
    the first group), the optimiser's slices start 16-byte aligned, and the sums are bit-identical to ONE flat all-reduce on both ranks."""
    world = 2
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_level_sync_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    f0, o0, n, offs = ret[0]
    f1, o1, _, _ = ret[1]
    assert np.array_equal(f0, f1)
    for cuts in o0:
        s0, groups, segments = o0[cuts]
        assert np.array_equal(s0, f0) and np.array_equal(o1[cuts][0], f0)
        assert len(groups) == len(cuts) + 1
        masks = [m for m, _, _ in groups]
        assert sum(masks) == (1 << 16) - 1 and all(a & b == 0 for i, a in enumerate(masks) for b in masks[i + 1:])
        spans = sorted((lo, hi) for _, lo, hi in groups)
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        # the optimiser's slices: 16-byte aligned starts, tiling the buffer too, each one inside what has arrived when its group was waited for
        segs = sorted(segments)
        assert segs[0][0] == 0 and segs[-1][1] == n and all(a[1] == b[0] for a, b in zip(segs, segs[1:])) and all(lo % 4 == 0 for lo, _ in segs)
        assert all(alo >= lo and alo - lo < 4 for (alo, _), (_, lo, _) in zip(segments, groups))
        assert groups[0][2] == n                                     # first group on the wire: the finest levels + the MLP weights
        for m, lo, hi in groups:                                     # a group's slice = its levels' rows (+ the tail for the first)
            first = min(l for l in range(16) if (m >> l) & 1)
            assert lo == offs[first] * 2
