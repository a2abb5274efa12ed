"""Pipeline-level data parallelism on the GPU: two ranks (gloo over 127.0.0.1, both on cuda:0 - the RCCL path needs
several GPUs and is the driver's to run) each run `NgpPipeline.train_step` on their shard of the rays, with the flat gradient
all-reduce, with the level-grouped exchange and with the sharded form (reduce-scatter, optimiser on 1/N, all-gather), and must end with the parameters a single process gets when it accumulates the two
shards' gradients itself and applies the optimiser with grad_scale 1/world - the averaging semantics of the reference's DDP
(common/trainer/basic_trainer.py:197-198)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_RAYS, STEPS = 1536, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(dev, mode='flat'):
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
    # (the level-grouped exchange belongs to the level-major step: 16 levels x 2 features, like the config)
    cfg = NgpConfig(n_levels=16 if mode == 'levels' else 8, hashmap_size=15, max_res=512, n_grid=64, n_sample=512, noise_std=0.0, lr=1e-2)
    fld = NgpField(cfg, device=dev, seed=3)
    fld.view('table').mul_(1000.0)
    pipe = NgpPipeline(fld, max_rays=2048, max_samples=1 << 17)
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.1, seed=5)))
    batches = []
    g = torch.Generator().manual_seed(11)
    for i in range(STEPS):
        o, d = synthetic_rays(N_RAYS, seed=40 + i, device=dev)
        batches.append((o, d, torch.rand(N_RAYS, 3, generator=g).to(dev), torch.rand(N_RAYS, 3, generator=g).to(dev)))
    return cfg, fld, pipe, batches


def _worker(rank, world, port, mode, path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from arcnerf_amd import distributed as D
    D.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    cfg, fld, pipe, batches = _setup(dev, mode)
    D.broadcast_params(fld.params, src=0)
    sync = D.ShardedGradSync(fld.n_params) if mode == 'sharded' else None
    if mode == 'levels':
        assert pipe.level_major
        sync = D.LevelGroupedGradSync(fld, (11, 5))
        assert len(sync.groups) == 3
    reduced = []

    def flat_all_reduce(t):
        D.allreduce_grads(t, world)
        if not reduced:
            reduced.append(t.clone())     # the summed gradient of the first step, as the optimiser sees it

    for o, d, tgt, bkg in batches:
        lo, hi = D.shard_range(N_RAYS, rank, world)
        pipe.rng.set_state(_rank_rng_state(pipe, rank))
        if sync is not None:
            pipe.train_step(o[lo:hi].contiguous(), d[lo:hi].contiguous(), tgt[lo:hi].contiguous(), bkg_color=bkg[lo:hi].contiguous(),
                            world_size=world, grad_sync=sync)
        else:
            pipe.train_step(o[lo:hi].contiguous(), d[lo:hi].contiguous(), tgt[lo:hi].contiguous(), bkg_color=bkg[lo:hi].contiguous(),
                            all_reduce=flat_all_reduce, world_size=world)
    torch.cuda.synchronize()
    np.save(path + '.rank{}.npy'.format(rank), fld.params.cpu().numpy())
    if reduced and rank == 0:
        np.save(path + '.grad0.npy', reduced[0].cpu().numpy())
    torch.distributed.destroy_process_group()


def _rank_rng_state(pipe, rank):
    """every rank marches with its own pcg32 stream (rank-local RNG like the reference): seed 9121 + rank, restarted per step so that
    the single-process replay below can reproduce it"""
    from arcnerf_amd.ops import functional as F
    return F.Pcg32Host(9121 + rank).state


@pytest.mark.parametrize('mode', ['flat', 'sharded', 'levels'])
def test_two_ranks_train_like_one_process_accumulating_both_shards(mode):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    world = 2
    ctx = mp.get_context('spawn')
    port = _free_port()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'params')
        procs = [ctx.Process(target=_worker, args=(r, world, port, mode, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        got = [np.load(path + '.rank{}.npy'.format(r)) for r in range(world)]
        grad0 = np.load(path + '.grad0.npy') if mode == 'flat' else None
    assert np.array_equal(got[0], got[1])           # replicas stay identical
    # single-process replay: both shards' gradients accumulated into the flat buffer, one optimiser pass with grad_scale 1/world
    from arcnerf_amd import distributed as D
    dev = torch.device('cuda:0')
    cfg, fld, pipe, batches = _setup(dev, mode)
    p0 = fld.params.cpu().numpy().copy()
    for step, (o, d, tgt, bkg) in enumerate(batches):
        fld.grads.zero_()
        for rank in range(world):
            lo, hi = D.shard_range(N_RAYS, rank, world)
            pipe.rng.set_state(_rank_rng_state(pipe, rank))
            oo, dd = o[lo:hi].contiguous(), d[lo:hi].contiguous()
            rgb, _, _ = pipe.forward(oo, dd, bkg[lo:hi].contiguous(), train=True)
            _, d_rgb = pipe.huber_grad(rgb, tgt[lo:hi].contiguous())
            pipe.backward(oo, dd, d_rgb)
        if step == 0 and grad0 is not None:
            # the collective is a SUM of the ranks' gradients (the 1/world of DDP's average is the optimiser's grad_scale): checked on
            # the gradient itself, because Adam's update is nearly invariant to a wrong scale
            ref_g = fld.grads.cpu().numpy()
            assert np.abs(grad0 - ref_g).max() <= 1e-5 * np.abs(ref_g).max()
        pipe.optimizer_step(world)
    torch.cuda.synchronize()
    ref = fld.params.cpu().numpy()
    moved = np.abs(ref - p0).max()
    assert moved > 1e-3                              # the steps did something
    # Adam (eps 1e-15) turns the summation-order noise of near-zero gradient entries into full-size steps of either sign: parameters
    # agree to a fraction of a percent of the distance they travelled, not bit for bit
    assert np.abs(got[0] - ref).max() <= 1e-2 * moved, (np.abs(got[0] - ref).max(), moved)


def _ddp_worker(rank, world, port, path):
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import train_ddp
    from arcnerf_amd import distributed as D
    from arcnerf_amd.pipeline import NgpConfig
    D.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    cfg = NgpConfig(hashmap_size=14, max_res=512, n_grid=32, n_sample=512, noise_std=0.0, lr=1e-2, epoch_optim=8, epoch_optim_warmup=16, white_bkg=True)
    out = train_ddp.train(cfg, dev, rank, world, steps=40, n_rays=1024, n_batches=4, sync='levels', balance=True, max_samples=1 << 17)
    torch.save(out, path + '.rank{}.pt'.format(rank))
    torch.distributed.destroy_process_group()


def test_two_ranks_train_with_applied_refreshes_and_balanced_shards():
    """tools/train_ddp.train on two ranks for 40 steps: the level-grouped gradient exchange, shards balanced by the per-ray sample counts of
    a batch's previous visit (distributed.balanced_shards), and FIVE applied occupancy refreshes (epoch_optim 8: one in the warm-up, four
    after it) each followed by the broadcast of rank 0's fields (distributed.broadcast_occupancy) - what DDP's broadcast_buffers does
    for the reference (common/trainer/basic_trainer.py:198).  Both ranks end with bit-identical parameters, bitfields and opacity fields;
    the occupancy was pruned; the shards tile the batch and moved away from the equal split once counts were known."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    world = 2
    ctx = mp.get_context('spawn')
    port = _free_port()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'ddp')
        procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, path)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        a, b = [torch.load(path + '.rank{}.pt'.format(r)) for r in range(world)]
    assert a['steps'] == b['steps'] == 40
    assert torch.equal(a['params'], b['params']) and torch.equal(a['bitfield'], b['bitfield']) and torch.equal(a['opafield'], b['opafield'])
    occ = float(a['bitfield'].float().mean())
    assert 0.0 < occ < 0.9, occ
    for (lo0, hi0), (lo1, hi1) in zip(a['shards'], b['shards']):
        assert lo0 == 0 and hi0 == lo1 and hi1 == 1024
    assert all(s == (0, 512) for s in a['shards'][:4]) and any(s != (0, 512) for s in a['shards'][4:])


def _run_bench(extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['ARCN_DIST_BACKEND'] = 'gloo'      # two ranks on ONE GPU: RCCL wants a GPU per rank, the launch path is the same
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + extra, env=env, cwd=root, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]      # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize('config', ['ngp', 'neus_ngp_multivol', 'ngp_module'])
def test_bench_gpus_flag_spawns_the_ranks_itself(config):
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form) must start two ranks itself (the reference:
    scripts/gpu.sh:9-21 -> basic_trainer.py:73-111 mp.spawn), report n_gpus = 2, the collective backend / world size it saw, the bytes
    per step of the gradient all-reduce, per-rank sample counts, and a whole-job value = all ranks' samples / max-over-ranks time."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    extra = ['--gpus', '2', '--steps', '4', '--warmup', '2', '--no-cpu-baseline']
    if config != 'ngp':
        extra += ['--config', config, '--rays', '1024']
    out = _run_bench(extra)
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'ray-sharded dp2' and out['scaling'] == 'weak'
    rc = out['rccl']
    assert rc['backend'] == 'gloo' and rc['world_size_seen'] == 2 and 'bench.py --gpus N' in rc['launcher']
    assert rc['allreduce_bytes_per_step'] >= 4 * out['config'].get('n_params', 12_000_000)
    assert len(rc['per_rank_samples_per_step']) == 2 and min(rc['per_rank_samples_per_step']) > 0
    total = sum(rc['per_rank_samples_per_step']) * out['steps']
    assert abs(out['value'] - total / (out['ms_per_step'] * 1e-3 * out['steps'])) <= 1e-3 * out['value']
    # weak scaling: both ranks carry a full batch (rank-local rays), so the job's samples are about twice one rank's
    assert 1.6 <= sum(rc['per_rank_samples_per_step']) / max(rc['per_rank_samples_per_step']) <= 2.0
    if config == 'neus_ngp_multivol':      # the hand-ordered chain ran on both ranks (not the module path), with its gradient exchange
        assert 'trainer.FusedNeusNgpStep' in out['config']['launch'] and 'gradient exchange: flat' in out['config']['launch']
        assert '({} steps'.format(out['steps'] + out['warmup']) in out['config']['launch']


def test_bench_refuses_more_rccl_ranks_than_gpus():
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'ARCN_DIST_BACKEND')}
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0'], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'one GPU per rank' in r.stderr


@pytest.mark.parametrize('config,sync', [('ngp', None), ('ngp', 'levels'), ('ngp', 'sharded'), ('neus_ngp_multivol', None)])
def test_bench_under_torchrun_on_a_one_rank_rccl_communicator(config, sync):
    """The driver's N > 1 command form (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    bench.py --gpus N`) with N = 1 and ARCN_DIST_FORCE=1: the process group is built on RCCL (backend "nccl"), and the gradient
    exchange (ARCN_GRAD_SYNC: flat = the default, levels, sharded), the parameter broadcast, the barriers around the timed region, the max-over-ranks reduction and the
    all-gather of the per-rank samples all run through a real RCCL communicator on this box's one GPU - everything of the multi-GPU
    path except a second rank."""
    import json
    import socket
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'ARCN_DIST_BACKEND'):
        env.pop(k, None)
    env.update({'ARCN_DIST_FORCE': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    env.pop('ARCN_GRAD_SEGMENTS', None)
    env.pop('ARCN_GRAD_SYNC', None)
    if sync is not None:        # None: the default of the N > 1 step, ONE flat all-reduce (north_star's form)
        env['ARCN_GRAD_SYNC'] = sync
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    extra = ['--gpus', '1', '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-other-configs', '--no-psnr']
    if config != 'ngp':
        extra += ['--config', config, '--rays', '1024']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py')] + extra
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    rc = out['rccl']
    assert out['n_gpus'] == 1 and rc is not None and rc['backend'] == 'nccl' and rc['world_size_seen'] == 1
    assert rc['allreduce_alone_ms'] > 0 and out['value'] > 0
    if config == 'ngp':
        assert rc['collectives_per_step'] in {None: (1,), 'levels': (2,), 'sharded': (2, 3)}[sync]      # (sharded: + the tail's all-reduce when n % 4 N != 0)
        assert rc['exposed_ms'] >= 0.0 and ('level groups' in rc['grad_sync']) == (sync == 'levels') and ('reduce-scatter' in rc['grad_sync']) == (sync == 'sharded')
        # a one-rank SUM is the identity: the step trains like the single-GPU step (two-pass optimiser form)
        assert 1e8 < out['value'] < 1e9


# ---- the drop-in module API, data parallel ---------------------------------------------------------------------------------------------
def _module_ddp_worker(rank, world, port, path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import g26_utils as U
    import torch.distributed as dist
    from arcnerf_amd import distributed as D
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    D.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ov = ['--model.rays.noise_std', '0.0', '--model.obj_bound.volume.n_grid', '32', '--model.obj_bound.epoch_optim', '4',
          '--model.obj_bound.epoch_optim_warmup', '8']
    lc = type('C', (), {})()
    lc.loss = type('C', (), {})()
    lc.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    R, half, steps = 512, 256, 10
    out = {}
    for mode in ('eager', 'fused'):
        torch.manual_seed(5)
        m = build_model(load_configs(os.path.join(root, 'configs', 'nerf_ngp.yaml'), ov)).to(dev)
        with torch.no_grad():
            m.fg_model.coarse_geo_net.embed_fn.embeddings.mul_(1000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15, weight_decay=1e-6, ema_decay=0.95, ema_in_param=True).flatten()
        opt.grad_scale = 1.0 / world
        D.broadcast_params(opt.flat_params(), src=0)
        ema = T.EMA(m, 0.95, opt)
        lf = T.build_loss(lc)
        sampler_rng(reset=True)
        m.train()
        stepper = T.FusedNgpStep(m, lf, opt, ema, world_size=world, grad_sync='levels', grad_level_cuts=(11, 5)) if mode == 'fused' else None
        losses = []
        for k in range(steps):
            m.optimize(k)
            inp = U.step_inputs(k, R)
            sl = slice(rank * half, (rank + 1) * half)
            feed = {'rays_o': torch.from_numpy(inp['rays_o'][sl])[None].to(dev), 'rays_d': torch.from_numpy(inp['rays_d'][sl])[None].to(dev),
                    'rays_r': torch.zeros(1, half, 1, device=dev), 'img': torch.from_numpy(inp['img'][sl])[None].to(dev),
                    'bkg_color': torch.from_numpy(inp['bkg_color'][sl])[None].to(dev)}
            if stepper is not None:
                _, loss = stepper(feed, k)
            else:       # the module path the way DistributedDataParallel runs it: backward, SUM of the flat gradient, optimiser with 1 / world
                o = m(feed, cur_epoch=k)
                loss = lf(feed, o)
                opt.zero_grad()
                loss['sum'].backward()
                dist.all_reduce(opt.flat_grads())
                opt.step()
                ema.ema_step()
            losses.append(float(loss['sum']))
        torch.cuda.synchronize()
        out[mode + '_params'] = opt.flat_params().cpu().numpy()
        out[mode + '_losses'] = np.array(losses)
        out[mode + '_bits'] = m.fg_model.obj_bound.volume.get_voxel_bitfield(flatten=True).cpu().numpy()
        if stepper is not None:
            out['fused_steps'] = np.array(stepper.steps)
            out['groups'] = np.array(len(stepper._sync.groups))
    np.savez(path + '.rank{}'.format(rank), **out)
    dist.destroy_process_group()


def test_two_ranks_through_the_module_api_fused_step_equal_the_eager_ddp_form():
    """trainer.FusedNgpStep(world_size=2): every rank its shard of the rays through NgpPipeline.train_step on the flattened optimiser's buffers,
    the gradient summed in level groups overlapped with the scatter, FusedAdam.grad_scale = 1 / 2, a refreshed occupancy broadcast from rank 0 -
    against the same two ranks on the module path with ONE all-reduce of the flat gradient between backward and FusedAdam.step (what the
    reference's DistributedDataParallel amounts to, common/trainer/basic_trainer.py:197-198).  Both forms leave bit-identical parameters on
    the two ranks; the two forms agree to the float scatter's order noise; ten steps, two applied refreshes."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'out')
        mp.spawn(_module_ddp_worker, args=(2, _free_port(), path), nprocs=2, join=True)
        r0, r1 = dict(np.load(path + '.rank0.npz')), dict(np.load(path + '.rank1.npz'))
    for mode in ('eager', 'fused'):
        assert np.array_equal(r0[mode + '_params'], r1[mode + '_params']), mode
        assert np.array_equal(r0[mode + '_bits'], r1[mode + '_bits']) and 0.0 < r0[mode + '_bits'].mean() < 1.0
    assert int(r0['fused_steps']) == 8 and int(r0['groups']) == 3
    assert np.array_equal(r0['eager_bits'], r0['fused_bits'])
    a, b = r0['eager_params'], r0['fused_params']
    far = np.abs(a - b) > 1e-3 * np.abs(a).max()
    assert far.mean() < 1e-3, far.mean()
    for r in (r0, r1):
        assert np.max(np.abs(r['eager_losses'] - r['fused_losses']) / np.abs(r['eager_losses'])) < 1e-4, (r['eager_losses'], r['fused_losses'])


# ---- config 4 (NeuS on the hash grid + MultiVol background), data parallel ------------------------------------------------------------------
def _neus_ddp_worker(rank, world, port, path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    from arcnerf_amd import distributed as D
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs
    D.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    loss_cfg = dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}})
    R, steps = 512, 4
    pool = []
    g = torch.Generator().manual_seed(3)
    for i in range(steps):
        o, d = synthetic_rays(2 * R, seed=20 + i, device=dev, radius=2.2)
        bk, im = torch.rand(1, 2 * R, 3, generator=g).to(dev), torch.rand(1, 2 * R, 3, generator=g).to(dev)
        sl = slice(rank * R, (rank + 1) * R)      # every rank its half of the global batch
        pool.append({'rays_o': o[sl].contiguous().view(1, -1, 3), 'rays_d': d[sl].contiguous().view(1, -1, 3), 'rays_r': torch.zeros(1, R, 1, device=dev),
                     'bkg_color': bk[:, sl].contiguous(), 'img': im[:, sl].contiguous()})
    out = {}
    for mode in ('eager', 'flat', 'sharded'):
        torch.manual_seed(0)
        m = build_model(load_configs(os.path.join(root, 'configs', 'neus_ngp_multivol.yaml'), [])).to(dev)
        m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
        m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(dev))
        with torch.no_grad():
            m.fg_model.geo_net.embed_fn.embeddings.mul_(200.0)
            m.bkg_model.geo_net.embed_fn.embeddings.mul_(2000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()
        opt.grad_scale = 1.0 / world
        D.broadcast_params(opt.flat_params(), src=0)
        lf = T.build_loss(loss_cfg)
        sampler_rng(reset=True)
        multivol_rng(reset=True)
        m.train()
        st = T.FusedNeusNgpStep(m, lf, opt, world_size=world, grad_sync=mode) if mode != 'eager' else None
        if st is not None:
            assert st.dist_step and not st.fuse_adam
        losses = []
        for k in range(steps):
            feed = pool[k]
            if st is not None:
                _, loss = st(feed, 20000 + k, next_feed_in=pool[k + 1] if k + 1 < steps else None)
            else:       # the module path the way DistributedDataParallel runs it: backward, SUM of the flat gradient, optimiser with 1 / world
                o_ = m(dict(feed), inference_only=False, cur_epoch=20000 + k)
                loss = lf(feed, o_)
                opt.zero_grad()
                loss['sum'].backward()
                dist.all_reduce(opt.flat_grads())
                opt.step()
            losses.append(float(loss['sum']))
        torch.cuda.synchronize()
        out[mode + '_params'] = opt.flat_params().cpu().numpy()
        out[mode + '_losses'] = np.array(losses)
        if st is not None:
            out[mode + '_steps'] = np.array(st.steps)
    np.savez(path + '.rank{}'.format(rank), **out)
    dist.destroy_process_group()


def test_two_ranks_through_the_fused_neus_ngp_step_equal_the_eager_ddp_form():
    """trainer.FusedNeusNgpStep(world_size=2): BASELINE config 4's hand-ordered chain on every rank's shard of the rays, the flat gradient summed
    between the scatters and the optimiser (one all-reduce | reduce-scatter + sharded Adam + all-gather), FusedAdam.grad_scale = 1 / 2 - against
    the same two ranks on the module path with ONE all-reduce of the flat gradient between backward and FusedAdam.step (what the reference's
    DistributedDataParallel amounts to, common/trainer/basic_trainer.py:192-198).  Every form leaves bit-identical parameters on the two
    ranks; the forms agree to the float scatter's order noise."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'out')
        mp.spawn(_neus_ddp_worker, args=(2, _free_port(), path), nprocs=2, join=True)
        r0, r1 = dict(np.load(path + '.rank0.npz')), dict(np.load(path + '.rank1.npz'))
    for mode in ('eager', 'flat', 'sharded'):
        assert np.array_equal(r0[mode + '_params'], r1[mode + '_params']), mode
    assert int(r0['flat_steps']) == 4 and int(r0['sharded_steps']) == 4
    a = r0['eager_params']
    for mode in ('flat', 'sharded'):
        b = r0[mode + '_params']
        far = np.abs(a - b) > 1e-3 * np.abs(a).max()
        assert far.mean() < 1e-3, (mode, far.mean())
        for r in (r0, r1):
            assert np.max(np.abs(r['eager_losses'] - r[mode + '_losses']) / np.abs(r['eager_losses'])) < 1e-4, (mode, r['eager_losses'], r[mode + '_losses'])
    assert not np.array_equal(r0['eager_losses'], r1['eager_losses'])      # (the ranks saw different rays)
