"""The instant-ngp training LOOP against a run of the reference's own loop (golden G26, tests/golden/make_golden_trajectory.py: the
reference's build_model + torch.optim.Adam from create_optimizer + its EMA class + ImgLoss Huber + VolumeBound.optimize +
Pipeline.fetch_step_update_dynamic_bs for 2 x 20 steps, K2 / K3 / K4 on the oracle, every random draw fed from tests/g26_utils.py).

Four reproductions, each over both legs (`a`: fresh start, warm-up refresh + post-warm-up refreshes, EMA de-bias from 0; `b`: a job
started at epoch 496: EMA n_step 496 with Adam at step 1, `epoch > 500` rule of the dynamic batch size):
  1. the module path: build_model(configs/nerf_ngp.yaml + the fixture's overrides) + trainer.train_epoch + FusedAdam (fused EMA);
  2. NgpPipeline.train_step / update_occupancy with the reference's net semantics (geometry output = [sigma | 15 features]): the
     scatter-fused optimiser of the single-GPU step;
  3. NgpPipeline with the config's FUSED nets (the bench's path: fused glue, step tail) against oracle/ngp_trainer.py, the CPU
     restatement of the loop that tests/test_oracle_trajectory_golden.py pins to the same fixture;
  4. the drop-in API at full speed: build_model(configs/nerf_ngp.yaml, fused nets) + trainer.train_epoch + trainer.FusedNgpStep (two batches
     in flight, VolumeBound.optimize of the module) against the same oracle loop as 3.

Bars (g26_utils.loss_bars): losses within 1e-4 relative until a refresh has had cells within 1e-4 of its threshold to decide (the threshold is the MEAN
opacity; the reference's own two runs decide them differently), from then on 5 x the distance between the reference's two runs; sample
counts equal while the bitfields are; bitfields equal except on the fixture's near-threshold cells; parameters through per-level sums,
stored rows and the full MLP weights at steps 1 / 4 / 8 / 12 / 20."""
import os

import numpy as np
import pytest
import torch

import g26_utils as U
from conftest import ROOT
from test_oracle_trajectory_golden import flat_params, make_cfg, nets_of

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, 'configs')
REPORT = os.environ.get('ARCN_TRAJ_REPORT') == '1'


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture()
def tape():
    import arcnerf_amd.geometry.volume as V
    V.set_refresh_tape(U.Tape())
    yield
    V.set_refresh_tape(None)


class Checker:
    """the per-step / per-refresh comparisons shared by the three reproductions; failures are collected and raised at the end of the run so
    that one run shows the whole trajectory (ARCN_TRAJ_REPORT=1 prints it)"""

    def __init__(self, g, leg, ref=None):
        self.g, self.leg, self.ref = g, leg, ref         # ref: per-step dicts of an oracle run instead of the fixture
        self.bars, self.cbars = U.loss_bars(g, leg), U.count_bars(g, leg)
        self.flips, self.n_ref, self.log, self.failures = 0, 0, [], []

    def expect(self, ok, *what):
        if not ok:
            self.failures.append(what)

    def refresh(self, k, refreshed, bits):
        g, leg = self.g, self.leg
        self.expect(int(refreshed) == int(g[leg + '_refreshed'][k]), leg, k, 'refresh cadence')
        if refreshed:
            try:
                if self.ref is None:
                    n = U.check_bitfield(g, leg, self.n_ref, bits, self.flips)
                else:       # an oracle run: its own bitfield and its own near-threshold cells
                    diff = np.asarray(bits, bool).reshape(-1) != self.ref['bitfields'][self.n_ref]
                    far = diff & ~self.ref['near'][self.n_ref]
                    # (a handful of cells may differ away from the band: Adam at eps 1e-15 gives a row whose gradient is rounding noise a step of
                    # size lr in either direction, and a refresh point that touches such a row moves that cell's opacity by ~1e-3)
                    assert self.flips > 0 or int(far.sum()) <= 4, (leg, 'refresh', self.n_ref, 'cells decided differently away from the threshold:', int(far.sum()))
                    assert int(diff.sum()) <= 0.02 * diff.size, (leg, 'refresh', self.n_ref, int(diff.sum()))
                    n = int(diff.sum())
                self.flips += n
                self.log.append(('refresh', self.n_ref, 'cells decided differently', n))
            except AssertionError as e:
                self.failures.append(e.args)
            self.n_ref += 1

    def step(self, k, n_rays, n_valid, loss):
        g, leg = self.g, self.leg
        want_rays = int(g[leg + '_n_rays'][k]) if self.ref is None else self.ref['n_rays'][k]
        want_valid = int(g[leg + '_n_valid'][k]) if self.ref is None else self.ref['n_valid'][k]
        want_loss = float(g[leg + '_loss'][k]) if self.ref is None else self.ref['loss'][k]
        self.expect(n_rays == want_rays, leg, 'step', k + 1, 'rays', n_rays, want_rays)
        if self.flips == 0:
            self.expect(n_valid == want_valid, leg, 'step', k + 1, 'samples', n_valid, want_valid)
        else:
            self.expect(abs(n_valid - want_valid) <= self.cbars[k] * want_valid, leg, 'step', k + 1, 'samples', n_valid, want_valid, float(self.cbars[k]))
        rel = abs(loss - want_loss) / want_loss
        bar = self.bars[k] if self.flips == 0 or self.ref is None else max(self.bars[k], 5e-3)
        self.log.append((k + 1, 'loss rel', rel, 'bar', float(bar), 'samples', n_valid, want_valid, 'flips so far', self.flips))
        self.expect(rel <= bar, leg, 'step', k + 1, 'loss', loss, want_loss, rel, float(bar), 'flips so far', self.flips)

    def params(self, k, tbl, nets):
        if (k + 1) not in U.SUMMARY_STEPS or self.ref is not None:
            return
        rep = U.param_report(self.g, self.leg, k + 1, tbl, nets)
        tight = self.bars[k] <= 1e-4 and self.flips == 0
        self.log.append(('params', k + 1, rep))
        try:
            U.check_params(rep, tight, (self.leg, k + 1))
        except AssertionError as e:
            self.failures.append(e.args)

    def done(self):
        self.expect(self.n_ref == (len(self.g[self.leg + '_bitfields']) if self.ref is None else len(self.ref['bitfields'])), 'number of refreshes')
        if REPORT:
            print('\n'.join(str(x) for x in self.log))
        assert not self.failures, self.failures


# ---- 1. the module path --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('leg', ['a', 'b'])
def test_module_path_reproduces_reference_loop(gpu, tape, leg):
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = U.golden()
    ov = [str(v) for v in g['overrides']] + ['--model.rays.noise_std', '0.0', '--model.rays.white_bkg', 'True']   # (the expr yaml's model block)
    cfgs = load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)
    m = build_model(cfgs).to(gpu)
    fg = m.fg_model
    assert not fg.packed_path_eligible()
    sd = {k[len(leg) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(leg + '_sd.')}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(('embeddings', 'bitfield', 'opafield')) for k in missing), (missing, unexpected)
    emb = fg.coarse_geo_net.embed_fn
    with torch.no_grad():
        emb.embeddings.copy_(torch.from_numpy(U.table(g, leg, emb.embeddings.shape[0])))
    lr, eps, wd, decay = [float(v) for v in g['optim']]
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=lr, eps=eps, weight_decay=wd, ema_decay=decay).flatten()
    ema = T.EMA(m, decay, opt)
    epochs = U.LEGS[leg]['epochs']
    ema.set_n_step(epochs[0])                      # ArcNerfTrainer.__init__ with progress.start_epoch
    loss_cfg = type('C', (), {})()
    loss_cfg.loss = type('C', (), {})()
    loss_cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=float(g['loss_cfg'][0]), weight=float(g['loss_cfg'][1])))()
    loss_factory = T.build_loss(loss_cfg)
    pipe = T.Pipeline()
    pipe.set_info('n_rays', U.N_RAYS0)
    pipe.set_info('dynamic_batch_size', U.UPDATE_EPOCH)
    pipe.set_info('dynamic_max_batch_size', U.N_RAYS_MAX)
    sampler_rng(reset=True)
    m.train()
    chk = Checker(g, leg)
    vol = fg.obj_bound.volume
    for k, epoch in enumerate(epochs):
        before = vol.get_voxel_opafield(flatten=True).clone()
        m.optimize(epoch)
        refreshed = not torch.equal(before, vol.get_voxel_opafield(flatten=True))
        chk.refresh(k, refreshed, vol.get_voxel_bitfield(flatten=True).cpu().numpy())
        n_rays = pipe.fetch_step_update_dynamic_bs(epoch, m)
        inp = U.step_inputs(epoch, n_rays)
        feed_in = {'rays_o': torch.from_numpy(inp['rays_o'])[None].to(gpu), 'rays_d': torch.from_numpy(inp['rays_d'])[None].to(gpu),
                   'rays_r': torch.zeros(1, n_rays, 1, device=gpu), 'img': torch.from_numpy(inp['img'])[None].to(gpu),
                   'bkg_color': torch.from_numpy(inp['bkg_color'])[None].to(gpu)}
        out, loss = T.step_optimize(m, feed_in, loss_factory, opt, ema, epoch)
        n_valid = int(fg._meter()._ring[fg._meter()._pending - 1]) if fg._meter()._pending else -1
        chk.step(k, n_rays, n_valid, float(loss['sum']))
        nets = {n: p.detach().cpu().numpy() for n, p in m.named_parameters() if p.requires_grad and not n.endswith('embeddings')}
        chk.params(k, emb.embeddings.detach().cpu().numpy(), nets)
    chk.done()


# ---- 2. / 3. NgpPipeline ---------------------------------------------------------------------------------------------------------------
def run_pipeline(gpu, cfg, flat, leg, chk, oracle_nets=False):
    from arcnerf_amd import trainer as T
    from arcnerf_amd.pipeline import NgpField, NgpPipeline
    fld = NgpField(cfg, device=gpu, seed=0)
    fld.params.copy_(torch.from_numpy(flat))
    pipe = NgpPipeline(fld, max_rays=U.N_RAYS_MAX, max_samples=1 << 19, prefetch_depth=1)
    epochs = U.LEGS[leg]['epochs']
    pipe.set_ema_n_step(epochs[0])
    meter = T.DynamicBsMeter(1 << U.LOG_MAX_ALLOWANCE)
    tp = T.Pipeline()
    tp.set_info('n_rays', U.N_RAYS0)
    tp.set_info('dynamic_batch_size', U.UPDATE_EPOCH)
    tp.set_info('dynamic_max_batch_size', U.N_RAYS_MAX)
    for k, epoch in enumerate(epochs):
        before = pipe.opafield.clone()
        pipe.update_occupancy(epoch, apply=True)
        refreshed = not torch.equal(before, pipe.opafield)
        chk.refresh(k, refreshed, pipe.bitfield.cpu().numpy())
        n_rays = tp.fetch_step_update_dynamic_bs(epoch, meter)
        inp = U.step_inputs(epoch, n_rays)
        o, d = torch.from_numpy(inp['rays_o']).to(gpu), torch.from_numpy(inp['rays_d']).to(gpu)
        loss = pipe.train_step(o, d, torch.from_numpy(inp['img']).to(gpu), bkg_color=torch.from_numpy(inp['bkg_color']).to(gpu))
        n_valid = pipe.sample_count()
        if n_valid > 0:
            meter.add(pipe.n_dev)
        chk.step(k, n_rays, n_valid, float(loss))
        p = fld.params.cpu().numpy()
        chk.params(k, p[fld._seg['table'][0]:fld._seg['table'][0] + fld._seg['table'][1]].reshape(-1, 2), nets_of(fld, p))
    chk.done()
    return pipe


@pytest.mark.parametrize('leg', ['a', 'b'])
def test_pipeline_train_step_reproduces_reference_loop(gpu, tape, leg):
    from arcnerf_amd.pipeline import NgpField
    g = U.golden()
    cfg = make_cfg()
    flat = flat_params(g, leg, NgpField(cfg, device='cpu', seed=0))
    pipe = run_pipeline(gpu, cfg, flat, leg, Checker(g, leg))
    assert pipe._adam_rest is not None, 'the single-GPU step is expected to run the scatter-fused optimiser'


def oracle_fused_run(oracle, g, leg):
    """the leg's start state for the config's FUSED nets (seeded, torch.nn.Linear's range, density row x the leg's scale) and the run of
    oracle/ngp_trainer.py from it: -> (cfg, flat, ref, trainer)"""
    from arcnerf_amd.pipeline import NgpField
    from oracle.ngp_trainer import OracleNgpTrainer
    cfg = make_cfg(geo_fused_semantics=True, W_feat=16)
    fld = NgpField(cfg, device='cpu', seed=0)
    rng = np.random.default_rng(2690)
    flat = fld.params.numpy().copy()
    off, n = fld._seg['table']
    flat[off:off + n] = U.table(g, leg, fld.offsets[-1]).reshape(-1)
    for name in ('geo_w', 'rad_w'):
        off, n = fld._seg[name]
        flat[off:off + n] = ((rng.random(n, dtype=np.float32) * 2 - 1) * np.float32(0.125)).astype(np.float32)
    off, _ = fld._seg['geo_w']
    w1 = off + fld.geo_dims[0] * fld.geo_dims[1]
    flat[w1:w1 + fld.geo_dims[1]] *= np.float32(U.LEGS[leg]['sigma_row_scale'])
    epochs = U.LEGS[leg]['epochs']
    tr = OracleNgpTrainer(oracle, fld, cfg, flat, 1 << U.LOG_MAX_ALLOWANCE, U.N_RAYS0, U.UPDATE_EPOCH, U.N_RAYS_MAX, start_epoch=epochs[0]).start_ema()
    ref = {'bitfields': [], 'near': [], 'n_rays': [], 'n_valid': [], 'loss': []}
    for epoch in epochs:
        perm, uni = U.refresh_draws(epoch, cfg.n_grid ** 3)
        if tr.optimize(epoch, perm, uni):
            ref['bitfields'].append(tr.bitfield.copy())
            ref['near'].append(np.abs(tr.opafield - tr.last['thres']) <= U.NEAR_BAND * tr.last['thres'])
        n_rays = tr.update_n_rays(epoch)
        inp = U.step_inputs(epoch, n_rays)
        res = tr.step(inp['rays_o'], inp['rays_d'], inp['bkg_color'], inp['img'])
        ref['n_rays'].append(n_rays)
        ref['n_valid'].append(res['n_samples'])
        ref['loss'].append(res['loss'])
    return cfg, flat, ref, tr


@pytest.mark.parametrize('leg', ['a', 'b'])
def test_fused_net_pipeline_follows_the_oracle_loop(gpu, tape, oracle, leg):
    """the bench's path (fused 32 -> 64 -> 16 geometry net whose whole output feeds the radiance net, fused glue, step tail): no CPU run of
    the reference exists for it (tiny-cuda-nn), so the trajectory comes from oracle/ngp_trainer.py - pinned to G26 for the reference's
    semantics by tests/test_oracle_trajectory_golden.py - switched to the fused semantics"""
    g = U.golden()
    cfg, flat, ref, tr = oracle_fused_run(oracle, g, leg)
    pipe = run_pipeline(gpu, cfg, flat, leg, Checker(g, leg, ref))
    assert pipe._tail is not None and pipe.fused_glue and pipe.level_major, 'expected the bench configuration of the step'
    # end state: the oracle's parameters, through the same summary
    p = pipe.field.params.cpu().numpy()
    la = np.abs(tr.p).astype(np.float64).sum()
    assert abs(np.abs(p).astype(np.float64).sum() - la) <= 2e-2 * la


@pytest.mark.parametrize('leg', ['a', 'b'])
def test_fused_module_step_follows_the_oracle_loop(gpu, tape, oracle, leg):
    """4. the drop-in API at full speed - build_model(configs/nerf_ngp.yaml, the fused nets UNCHANGED) + FusedAdam(ema_in_param).flatten() +
    trainer.train_epoch with trainer.FusedNgpStep (two batches in flight, the refresh through the module's own VolumeBound.optimize) -
    against the same oracle loop as 3., from the same start state (the flattened optimiser's buffer IS the pipeline's flat layout)"""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = U.golden()
    cfg, flat, ref, tr = oracle_fused_run(oracle, g, leg)
    ov = ['--model.rays.noise_std', '0.0', '--model.rays.white_bkg', 'True', '--model.obj_bound.volume.n_grid', str(U.N_GRID),
          '--model.obj_bound.epoch_optim', str(U.EPOCH_OPTIM), '--model.obj_bound.epoch_optim_warmup', str(U.EPOCH_WARMUP),
          '--model.obj_bound.log_max_allowance', str(U.LOG_MAX_ALLOWANCE)]
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)).to(gpu)
    fg = m.fg_model
    assert fg.packed_path_eligible() and fg.get_n_coarse_sample() == cfg.n_sample
    lr, eps, wd, decay = [float(v) for v in g['optim']]
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=lr, eps=eps, weight_decay=wd, ema_decay=decay, ema_in_param=True).flatten()
    assert opt.flat_params().numel() == flat.shape[0]
    with torch.no_grad():
        opt.flat_params().copy_(torch.from_numpy(flat))
    ema = T.EMA(m, decay, opt)
    epochs = U.LEGS[leg]['epochs']
    ema.set_n_step(epochs[0])
    loss_cfg = type('C', (), {})()
    loss_cfg.loss = type('C', (), {})()
    loss_cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=float(g['loss_cfg'][0]), weight=float(g['loss_cfg'][1])))()
    loss_factory = T.build_loss(loss_cfg)
    tp = T.Pipeline()
    tp.set_info('n_rays', U.N_RAYS0)
    tp.set_info('dynamic_batch_size', U.UPDATE_EPOCH)
    tp.set_info('dynamic_max_batch_size', U.N_RAYS_MAX)
    sampler_rng(reset=True)
    m.train()
    stepper = T.FusedNgpStep(m, loss_factory, opt, ema, max_rays=U.N_RAYS_MAX)
    drawn = []

    def get_batch(n_rays):
        epoch = epochs[len(drawn)]
        drawn.append(n_rays)
        inp = U.step_inputs(epoch, n_rays)
        return {'rays_o': torch.from_numpy(inp['rays_o'])[None].to(gpu), 'rays_d': torch.from_numpy(inp['rays_d'])[None].to(gpu),
                'img': torch.from_numpy(inp['img'])[None].to(gpu), 'bkg_color': torch.from_numpy(inp['bkg_color'])[None].to(gpu)}

    chk = Checker(g, leg, ref)
    vol = fg.obj_bound.volume
    for k, epoch in enumerate(epochs):
        before = vol.get_voxel_opafield(flatten=True).clone()
        out, loss = T.train_epoch(m, get_batch, loss_factory, opt, ema, tp, epoch, total_epoch=epochs[-1] + 1, stepper=stepper)
        refreshed = not torch.equal(before, vol.get_voxel_opafield(flatten=True))
        chk.refresh(k, refreshed, vol.get_voxel_bitfield(flatten=True).cpu().numpy())
        pipe = stepper.pipe if stepper.pipe is not None and stepper.steps > 0 else fg._pipe
        chk.step(k, drawn[k], int(pipe.n_dev.item()), float(loss['sum']))
    chk.done()
    assert stepper.steps == len(epochs) - 2 and drawn == ref['n_rays']
    p = opt.flat_params().cpu().numpy()
    la = np.abs(tr.p).astype(np.float64).sum()
    assert abs(np.abs(p).astype(np.float64).sum() - la) <= 2e-2 * la


# ---- the drop-in step on the pipeline's fused step ----------------------------------------------------------------------------------
def test_fused_module_step_equals_eager_steps(gpu):
    """trainer.FusedNgpStep runs `model(feed_in) -> ImgLoss -> backward -> FusedAdam.step (+ fused EMA)` of configs/nerf_ngp.yaml as
    NgpPipeline.train_step on the flattened optimiser's buffers (loss inside the compositor, optimiser inside the scatter, the next two
    batches marched early by trainer.train_epoch).  Against the same model trained through trainer.train_epoch's eager form, over
    epochs 496 .. 519: five occupancy refreshes (every 4 epochs: the step after each marches inline), four changes of the dynamic batch
    size (epoch > 500), equal ray counts, equal sample counts, the sampler's generator in the same state, losses to 1e-5, the
    parameters and the optimiser's counters at the end."""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    ov = ['--model.rays.noise_std', '0.0', '--model.obj_bound.volume.n_grid', '32', '--model.obj_bound.epoch_optim', '4',
          '--model.obj_bound.epoch_optim_warmup', '8']
    loss_cfg = type('C', (), {})()
    loss_cfg.loss = type('C', (), {})()
    loss_cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    epochs = list(range(496, 520))
    runs = {}
    for mode in ('eager', 'fused'):
        torch.manual_seed(5)
        m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)).to(gpu)
        fg = m.fg_model
        assert fg.packed_path_eligible()
        with torch.no_grad():
            fg.coarse_geo_net.embed_fn.embeddings.mul_(1000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15, weight_decay=1e-6, ema_decay=0.95, ema_in_param=True).flatten()
        ema = T.EMA(m, 0.95, opt)
        ema.set_n_step(epochs[0])
        loss_factory = T.build_loss(loss_cfg)
        tp = T.Pipeline()
        tp.set_info('n_rays', 256)
        tp.set_info('dynamic_batch_size', 4)
        tp.set_info('dynamic_max_batch_size', 1024)
        sampler_rng(reset=True)
        m.train()
        stepper = T.FusedNgpStep(m, loss_factory, opt, ema, max_rays=512) if mode == 'fused' else None
        drawn = []

        def get_batch(n_rays):
            k = epochs[0] + len(drawn)          # batches are drawn in epoch order, one per epoch (a step early or not)
            drawn.append(n_rays)
            inp = U.step_inputs(k, n_rays)
            return {'rays_o': torch.from_numpy(inp['rays_o'])[None].to(gpu), 'rays_d': torch.from_numpy(inp['rays_d'])[None].to(gpu),
                    'rays_r': torch.zeros(1, n_rays, 1, device=gpu), 'img': torch.from_numpy(inp['img'])[None].to(gpu),
                    'bkg_color': torch.from_numpy(inp['bkg_color'])[None].to(gpu)}

        losses, counts, ahead = [], [], 0
        for epoch in epochs:
            if stepper is not None and stepper._queue and stepper._queue[0][0] == epoch:
                ahead += 1
                if epoch == 510:    # buffers rebuilt while TWO batches marched ahead sit in the old ones: marched again, the same generator launches
                    assert len(stepper.pipe._prefetched) == 2 and [e for e, _ in stepper._queue] == [510, 511]
                    stepper._build(gpu, 1024, min_samples=stepper.pipe.cap + 1024)
            out, loss = T.train_epoch(m, get_batch, loss_factory, opt, ema, tp, epoch, total_epoch=epochs[-1] + 1, stepper=stepper)
            losses.append(float(loss['sum']))
            pipe = stepper.pipe if stepper is not None and stepper.pipe is not None else fg._pipe
            counts.append(int(pipe.n_dev.item()))
        torch.cuda.synchronize()
        runs[mode] = dict(losses=losses, counts=counts, drawn=drawn, params=opt.flat_params().clone(), rng=sampler_rng().state, step=opt._flat[0]['step'],
                          ema=(ema.n_step, opt.ema_n_step), occ=float(fg.obj_bound.volume.get_voxel_bitfield().float().mean()),
                          out={k: v.detach().clone() for k, v in out.items()}, grads=float(opt.flat_grads().abs().max()), ahead=ahead,
                          steps=stepper.steps if stepper is not None else 0, rebuilds=stepper.rebuilds if stepper is not None else 0)
    a, b = runs['eager'], runs['fused']
    assert b['steps'] == len(epochs) - 2 and b['rebuilds'] >= 3      # first build, 256 -> more rays than max_rays, the forced one
    # a batch is drawn a step early unless its epoch refreshes the occupancy (496 + 4 i) - the batch-size rule acts at the same epochs
    assert b['ahead'] == sum(1 for e in epochs[4:] if e % 4 != 0), b['ahead']
    assert a['drawn'] == b['drawn'] and len(set(a['drawn'])) >= 2, (a['drawn'], b['drawn'])
    assert a['counts'] == b['counts'], (a['counts'], b['counts'])
    assert a['rng'] == b['rng'] and a['step'] == b['step'] == len(epochs) and a['ema'] == b['ema'] == (epochs[-1] + 1, epochs[-1] + 1)
    assert 0.0 < a['occ'] < 1.0 and a['occ'] == b['occ']
    assert max(abs(x - y) / abs(x) for x, y in zip(a['losses'], b['losses'])) < 1e-5, (a['losses'], b['losses'])
    far = ((a['params'] - b['params']).abs() > 1e-3 * float(a['params'].abs().max())).float().mean()
    assert float(far) < 1e-3, float(far)
    assert b['grads'] == 0.0                     # the fused step leaves the flat gradient cleared
    for k in ('rgb_coarse', 'depth_coarse', 'mask_coarse'):
        assert a['out'][k].shape == b['out'][k].shape
        assert torch.allclose(a['out'][k], b['out'][k], rtol=1e-4, atol=1e-4), k


def test_fused_module_step_after_a_reflatten_and_around_a_progress_step(gpu):
    """FusedAdam.load_state_dict re-homes the flat buffers (the stepper binds the new ones); an iteration with get_progress=True runs on
    the module path (the per-sample outputs exist there only) with the fused steps continuing after it; gradient clipping is refused"""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.utils.cfgs_utils import load_configs
    ov = ['--model.rays.noise_std', '0.0', '--model.obj_bound.volume.n_grid', '32', '--model.obj_bound.epoch_optim', '4',
          '--model.obj_bound.epoch_optim_warmup', '8']
    loss_cfg = type('C', (), {})()
    loss_cfg.loss = type('C', (), {})()
    loss_cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    runs = {}
    for mode in ('eager', 'fused'):
        torch.manual_seed(5)
        m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)).to(gpu)
        with torch.no_grad():
            m.fg_model.coarse_geo_net.embed_fn.embeddings.mul_(1000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15, weight_decay=1e-6, ema_decay=0.95).flatten()   # (a separate EMA shadow)
        ema = T.EMA(m, 0.95, opt)
        loss_factory = T.build_loss(loss_cfg)
        sampler_rng(reset=True)
        m.train()
        stepper = T.FusedNgpStep(m, loss_factory, opt, ema) if mode == 'fused' else None
        if stepper is not None:
            with pytest.raises(RuntimeError, match='clipping'):
                T.FusedNgpStep(m, loss_factory, opt, ema, clip_value=0.1)
        losses = []
        for k in range(12):
            m.optimize(k)
            inp = U.step_inputs(k, 256)
            feed_in = {'rays_o': torch.from_numpy(inp['rays_o'])[None].to(gpu), 'rays_d': torch.from_numpy(inp['rays_d'])[None].to(gpu),
                       'rays_r': torch.zeros(1, 256, 1, device=gpu), 'img': torch.from_numpy(inp['img'])[None].to(gpu),
                       'bkg_color': torch.from_numpy(inp['bkg_color'])[None].to(gpu)}
            if k == 6:
                opt.load_state_dict(opt.state_dict())
            progress = k == 8
            if stepper is not None:
                out, loss = stepper(feed_in, k, get_progress=progress)
            else:
                out, loss = T.step_optimize(m, feed_in, loss_factory, opt, ema, k, get_progress=progress)
            if progress:
                assert any(key.startswith('progress') for key in out), list(out)
            losses.append(float(loss['sum']))
        torch.cuda.synchronize()
        runs[mode] = (losses, opt.flat_params().clone(), opt._flat[0]['step'], ema.n_step, stepper.steps if stepper else 0, stepper.rebuilds if stepper else 0)
    (la, pa, sa, ea, _, _), (lb, pb, sb, eb, n_fused, rebuilds) = runs['eager'], runs['fused']
    assert n_fused == 9 and rebuilds == 2 and sa == sb == 12 and ea == eb == 12
    assert max(abs(a - b) / abs(a) for a, b in zip(la, lb)) < 1e-5, (la, lb)
    far = ((pa - pb).abs() > 1e-3 * float(pa.abs().max())).float().mean()
    assert float(far) < 1e-3, float(far)
