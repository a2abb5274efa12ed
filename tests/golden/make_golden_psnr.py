"""G27 "PSNR@iter": the reference's TRAINING LOOP for the instant-ngp configuration WITH ITS OWN DATA PIPELINE, 600 iterations x 4 seeds on
a small analytic scene, held-out PSNR at checkpoints.  Run in the build container only.

What runs is the reference's own code, composed the way its trainer composes it (arcnerf/trainer/arcnerf_trainer.py:86-95 prepare_data,
:167-219 set_train_dataset / concat_train_batch, :494-548 train_epoch, :319-333 step_optimize, :555-571 train):

    pipeline = Pipeline(); pipeline.set_n_rays(n_rays); pipeline.setup_cfgs(cfgs.dataset.train.scheduler)
    data = {img (N, HW, 3), mask (N, HW), rays_o / rays_d (N, HW, 3), rays_r (N, HW, 1), H, W}       what concat_train_batch collects
    data = pipeline.process_train_data(logger, data)        centre precrop -> cross-view randperm shuffle -> dynamic batch size -> bkg colour
    for epoch:  model.optimize(epoch)                                               VolumeBound.optimize
                crop_shuffle / full_shuffle -> (the dataset again,) process_train_data      arcnerf_trainer.py:531-540
                batch = pipeline.get_train_batch(data, epoch, model)                dynamic bs, the next n_rays of the shuffle, random bkg blend
                feed_in = get_model_feed_in(batch); output = model(feed_in); loss; zero_grad; backward; Adam; ema.ema_step()
                at the checkpoints: model.eval(); model(held-out view, inference_only=True) -> PSNR on white (img_metric.py:50-56)

with `configs/expr/NeRF/lego/nerf_lego_nerf_ngp.yaml` (its model / optim / loss / dataset.train.scheduler blocks: precrop ratio 0.5, random
bkg colour, dynamic batch size; Adam 1e-1, eps 1e-15, weight decay 1e-6, EMA 0.95) and the size overrides of tests/g27_utils.py: 100 x 100
images (24 training views, 4 held out), a 32^3 grid refreshed every 8 iterations (warm-up 32), 256 samples per ray, 1024 rays per batch,
precrop.max_epoch 50, torch back-ends for the nets and the encoders as in G21 / G26 (`nb`).  The CUDA-only calls are the oracle's K2 / K3 /
K4 as in G21.  The dataset's rays are the reference's own get_rays (what Base3dDataset.precache_ray stores).  Every random draw of the
loop - torch.randperm in Pipeline.step_ray_sample, torch.rand_like in Pipeline.fetch_step_bkg_color, torch.randperm / torch.rand_like in
VolumeBound.optimize - is FED from tests/g27_utils.py (numpy PCG64) so that the mirror can be fed the same numbers.

A quirk the fixture pins (trainer/pipeline.py:95-118): the SECOND call of process_train_data clears `crop_max_epoch`; when a full pass over
the cropped rays ends before precrop.max_epoch (cropped rays / n_rays < max_epoch) the reshuffle is that second call and the crop never
ends.  The Lego recipe is on the safe side (16 M cropped rays / 4096 = 3906 > 500) and so are the training runs here (60 000 / 1024 =
58.6 > 50); the leg `quirk_*` is the data side alone (no model) with precrop.max_epoch 100 for 130 iterations: a pass ends at iteration
59, the reshuffle clears the crop's end, iteration 100 and everything after it still draws from the centre windows.

usage: python tests/golden/make_golden_psnr.py            (G27_EPOCHS / G27_SEEDS override the sizes for a dry run: no fixture is written)
"""
import os
import sys
import time
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, ROOT)
sys.path.insert(2, os.path.join(ROOT, 'tests'))
_r = types.ModuleType('pytorch3d.transforms.rotation_conversions')
for _n in ['axis_angle_to_matrix', 'matrix_to_axis_angle', 'matrix_to_rotation_6d', 'rotation_6d_to_matrix']:
    setattr(_r, _n, lambda *a, **k: None)
sys.modules['pytorch3d'] = types.ModuleType('pytorch3d')
sys.modules['pytorch3d.transforms'] = types.ModuleType('pytorch3d.transforms')
sys.modules['pytorch3d.transforms.rotation_conversions'] = _r
sys.modules.setdefault('cv2', types.ModuleType('cv2'))     # (imported by arcnerf.render.camera at module level; nothing here calls it)
import warnings  # noqa: E402

warnings.filterwarnings('ignore')
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arcnerf.geometry.volume as ref_volume  # noqa: E402
import arcnerf.models.base_modules.obj_bound.volume_bound as ref_vb  # noqa: E402
from arcnerf.datasets import get_model_feed_in  # noqa: E402
from arcnerf.loss import build_loss  # noqa: E402
from arcnerf.models import build_model  # noqa: E402
from arcnerf.render.ray_helper import get_rays  # noqa: E402
from arcnerf.trainer.ema import EMA  # noqa: E402
from arcnerf.trainer.pipeline import Pipeline  # noqa: E402
from common.trainer.optimizer import create_optimizer  # noqa: E402
from common.utils.cfgs_utils import get_value_from_cfgs_field, load_configs  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import g27_utils as U  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
EXPR = '/root/reference/configs/expr/NeRF/lego/nerf_lego_nerf_ngp.yaml'
OVERRIDES = ['--model.geometry.type', 'GeoNet', '--model.geometry.encoder.backend', 'torch', '--model.geometry.encoder.dtype', 'torch.float32',
             '--model.radiance.type', 'RadianceNet', '--model.radiance.encoder.view.backend', 'torch',
             '--model.geometry.use_bias', 'False', '--model.geometry.W_feat', '15', '--model.radiance.use_bias', 'False',
             '--model.radiance.W_feat_in', '15', '--model.chunk_pts', '4096',
             '--model.obj_bound.volume.n_grid', str(U.N_GRID), '--model.obj_bound.epoch_optim', str(U.EPOCH_OPTIM),
             '--model.obj_bound.epoch_optim_warmup', str(U.EPOCH_WARMUP), '--model.obj_bound.log_max_allowance', str(U.LOG_MAX_ALLOWANCE),
             '--model.rays.n_sample', str(U.N_SAMPLE), '--n_rays', str(U.N_RAYS0),
             '--dataset.train.scheduler.precrop.max_epoch', str(U.PRECROP_MAX_EPOCH),
             '--dataset.train.scheduler.dynamic_batch_size.update_epoch', str(U.UPDATE_EPOCH),
             '--dataset.train.scheduler.dynamic_batch_size.max_batch_size', str(U.N_RAYS_MAX)]
N_EPOCH = int(os.environ.get('G27_EPOCHS', U.N_EPOCH))
SEEDS = tuple(int(s) for s in os.environ['G27_SEEDS'].split(',')) if 'G27_SEEDS' in os.environ else U.SEEDS
DRY = N_EPOCH != U.N_EPOCH or SEEDS != U.SEEDS
_state = {}
_orig_aabb = ref_volume.aabb_ray_intersection


def oracle_k3(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance=0.0):
    h = _state['rng']
    aabb23 = aabb_range.permute(1, 0).contiguous().numpy()
    z, m, c = orc.sparse_volume_sampling(rays_o.numpy(), rays_d.numpy(), near.numpy(), far.numpy(), n_pts, np.float32(dt), aabb23, n_grid,
                                         bitfield.numpy(), near_distance, h.state, h.inc)
    h.advance()
    _state['n_valid'] = int(m.sum())
    return torch.from_numpy(z), torch.from_numpy(m)


def oracle_k2(rays_o, rays_d, aabb_range, eps=1e-7, force_torch=False):
    if force_torch:
        return _orig_aabb(rays_o, rays_d, aabb_range, eps, True)
    near, far, pts, mask = orc.aabb_intersection(rays_o.numpy(), rays_d.numpy(), aabb_range.permute(0, 2, 1).contiguous().numpy())
    return torch.from_numpy(near), torch.from_numpy(far), torch.from_numpy(pts), torch.from_numpy(mask)


def oracle_k4(full_tensor, group_idx, n_group):
    return torch.from_numpy(orc.tensor_reduce_max(full_tensor.numpy(), group_idx.numpy(), int(n_group)))


ref_vb.CUDA_BACKEND_AVAILABLE = True
ref_vb.sparse_volume_sampling = oracle_k3
ref_vb.tensor_reduce_max = oracle_k4
ref_volume.aabb_ray_intersection = oracle_k2


class Fed:
    """torch.randperm / torch.rand_like return what `perm(n)` / `uni(shape)` hand out while the block runs; `calls` records them"""

    def __init__(self, perm=None, uni=None):
        self.perm, self.uni, self.calls = perm, uni, []

    def __enter__(self):
        self._rp, self._rl = torch.randperm, torch.rand_like

        def randperm(n, **kw):
            self.calls.append(('perm', int(n)))
            return torch.from_numpy(self.perm(int(n)).copy())

        def rand_like(t, **kw):
            self.calls.append(('uni', tuple(t.shape)))
            return torch.from_numpy(self.uni(tuple(t.shape)).copy())
        torch.randperm, torch.rand_like = randperm, rand_like
        return self

    def __exit__(self, *a):
        torch.randperm, torch.rand_like = self._rp, self._rl


class Log:
    def __init__(self):
        self.lines = []

    def add_log(self, msg, level='info'):
        self.lines.append(msg)


def render_images():
    cache = '/tmp/g27_images.npz'
    if os.path.exists(cache):
        c = np.load(cache)
        return c['train'], c['test']
    t0 = time.time()
    tr = np.stack([U.render_view(*U.camera(v)) for v in range(U.TRAIN_VIEWS[0], U.TRAIN_VIEWS[0] + U.N_TRAIN)])
    te = np.stack([U.render_view(*U.camera(v)) for v in range(U.TEST_VIEWS[0], U.TEST_VIEWS[0] + U.N_TEST)])
    print('rendered {} + {} views in {:.0f} s'.format(len(tr), len(te), time.time() - t0), flush=True)
    np.savez_compressed(cache, train=tr, test=te)
    return tr, te


def view_rays(first, n):
    """the ray bundles Base3dDataset.precache_ray holds: cameras[i].get_rays(wh_order=False, center_pixel=True, normalize_rays_d=True)"""
    Ks, Ms = U.cameras(first, n)
    o, d, r = [], [], []
    for K, M in zip(Ks, Ms):
        ro, rd, _, rr = get_rays(U.W, U.H, torch.from_numpy(K), torch.from_numpy(M), wh_order=False, center_pixel=True, normalize_rays_d=True)
        o.append(ro), d.append(rd), r.append(rr)
    return torch.stack(o), torch.stack(d), torch.stack(r)


def run(seed, rgba_train, rgba_test, out, verbose=True):
    tag = 's{}_'.format(seed)
    cfgs = load_configs(EXPR, OVERRIDES)
    torch.manual_seed(2700 + seed)
    model = build_model(cfgs, None)
    fg = model.fg_model
    emb = fg.coarse_geo_net.embed_fn
    vol = fg.obj_bound.volume
    n_cells = U.N_GRID ** 3
    assert fg.get_ray_cfgs('n_sample') == U.N_SAMPLE and vol.get_n_grid() == U.N_GRID
    with torch.no_grad():
        emb.embeddings.copy_(torch.from_numpy(U.table_from_seed(emb.n_total_embed, emb.n_feat_per_entry, seed)))
    offsets = np.array(emb.offsets, np.int64)
    t0 = emb.embeddings.detach().numpy()
    out[tag + 'table_sum'], out[tag + 'table_probe'] = np.array(t0.astype(np.float64).sum()), t0[::100003].copy()
    for k, v in model.state_dict().items():
        if not k.endswith(('embed_fn.embeddings', '.volume_pts', '.grid_pts', '.corner')) and 'bitfield' not in k and 'opafield' not in k:
            out[tag + 'sd.' + k] = v.numpy().copy()
    if 'offsets' not in out:
        out['offsets'], out['resolutions'] = offsets, np.array(emb.resolutions, np.int64)
        out['overrides'] = np.array(OVERRIDES)
        out['optim'] = np.array([float(cfgs.optim.lr), float(cfgs.optim.eps), float(cfgs.optim.weight_decay), float(cfgs.optim.ema.decay)])
        out['loss_cfg'] = np.array([float(cfgs.loss.ImgLoss.delta), float(cfgs.loss.ImgLoss.weight)])
        sch = cfgs.dataset.train.scheduler
        out['scheduler'] = np.array([float(sch.precrop.ratio), float(sch.precrop.max_epoch), float(sch.dynamic_batch_size.update_epoch),
                                     float(sch.dynamic_batch_size.max_batch_size)])
        assert sch.bkg_color.color == 'random'
    params = [p for _, p in model.named_parameters()]
    optimizer = create_optimizer(parameters=params, **cfgs.optim.__dict__)
    loss_factory = build_loss(cfgs, None)
    ema = EMA(model, float(cfgs.optim.ema.decay))
    ema.set_n_step(0)
    log = Log()
    pipe = Pipeline()
    pipe.set_n_rays(log, get_value_from_cfgs_field(cfgs, 'n_rays', 1024))
    pipe.setup_cfgs(get_value_from_cfgs_field(cfgs.dataset.train, 'scheduler', None))

    img, mask = (torch.from_numpy(a) for a in U.dataset_tensors(rgba_train))
    rays_o, rays_d, rays_r = view_rays(U.TRAIN_VIEWS[0], U.N_TRAIN)

    def set_train_dataset():
        return {'img': img.clone(), 'mask': mask.clone(), 'rays_o': rays_o.clone(), 'rays_d': rays_d.clone(), 'rays_r': rays_r.clone(), 'H': U.H, 'W': U.W}

    n_shuffle = [0]

    def process(data):
        k = n_shuffle[0]
        with Fed(perm=lambda n: U.shuffle_perm(seed, k, n)) as fd:
            data = pipe.process_train_data(log, data)
        assert [c[0] for c in fd.calls] == ['perm']
        n_shuffle[0] += 1
        return data, fd.calls[0][1]

    test_o, test_d, test_r = view_rays(U.TEST_VIEWS[0], U.N_TEST)
    test_tgt = U.white_targets(rgba_test)

    def evaluate():
        model.eval()
        preds = []
        with torch.no_grad():
            for v in range(U.N_TEST):
                o = model({'rays_o': test_o[v][None], 'rays_d': test_d[v][None], 'rays_r': test_r[v][None]}, inference_only=True)
                preds.append(o['rgb'][0].numpy())
        model.train()
        preds = np.stack(preds)
        return U.psnr(preds, test_tgt), float(np.mean(1.0 - preds.min(-1) < 0.02))      # PSNR; share of pixels that are (almost) white

    _state['rng'] = orc.Pcg32(9121)
    model.train()
    data, total = process(set_train_dataset())
    rec = {k: [] for k in ('n_rays', 'loss', 'n_valid', 'refreshed', 'popcount', 'thres', 'n_near', 'n_refresh_pts', 'dyn_factor', 'shuffle_at', 'shuffle_total',
                           'batch_sums', 'psnr', 'white_share', 'occupied')}
    rec['shuffle_at'].append(-1), rec['shuffle_total'].append(total)
    bitfields, near = [], []
    t_start = time.time()
    for epoch in range(N_EPOCH):
        # ---- model.optimize(epoch), fed draws
        perm_uni = {}

        def r_perm(n):
            perm_uni['d'] = U.refresh_draws(seed, epoch, n_cells)
            assert n == n_cells
            return perm_uni['d'][0]

        def r_uni(shape):
            if 'd' not in perm_uni:
                perm_uni['d'] = U.refresh_draws(seed, epoch, n_cells)
            assert len(shape) == 2 and shape[1] == 3 and shape[0] <= n_cells
            return perm_uni['d'][1][:shape[0]]
        with Fed(perm=r_perm, uni=r_uni) as fd:
            model.optimize(epoch)
        refreshed = len(fd.calls) > 0
        rec['refreshed'].append(int(refreshed))
        if refreshed:
            opa = vol.get_voxel_opafield(flatten=True)
            thres = min(vol.get_mean_voxel_opacity(), fg.get_optim_cfgs('opa_thres'))
            bits = vol.get_voxel_bitfield(flatten=True).numpy()
            assert np.array_equal(bits, (opa >= thres).numpy())
            nr = ((opa - thres).abs() <= U.NEAR_BAND * thres).numpy()
            if len(bitfields) < U.N_KEEP_REFRESH:
                bitfields.append(np.packbits(bits, bitorder='little'))
                near.append(np.packbits(nr, bitorder='little'))
            rec['n_near'].append(int(nr.sum()))
            rec['popcount'].append(int(bits.sum()))
            rec['thres'].append(float(thres))
            rec['n_refresh_pts'].append(int([c for c in fd.calls if c[0] == 'uni'][0][1][0]))
        # ---- the data side of train_epoch (arcnerf_trainer.py:531-540)
        crop_shuffle, full_shuffle = pipe.check_crop_shuffle(epoch), pipe.check_full_shuffle()
        if crop_shuffle:
            data, total = process(set_train_dataset())
        elif full_shuffle:
            assert pipe.get_info('sample_cross_view')
            data, total = process(data)
        if crop_shuffle or full_shuffle:
            rec['shuffle_at'].append(epoch), rec['shuffle_total'].append(total)
        # ---- the batch
        cnt = fg.get_render_cfgs('measured_count')
        factor_peek = fg.get_render_cfgs('measured_batch_size') / cnt if cnt > 0 else 1.0
        with Fed(uni=lambda shape: U.bkg_draw(seed, epoch, shape[1])[None]) as fd:
            batch = pipe.get_train_batch(data, epoch, model)
        assert [c[0] for c in fd.calls] == ['uni'] and fd.calls[0][1][0] == 1
        applied = fg.get_render_cfgs('measured_count') == 0 and cnt > 0
        rec['dyn_factor'].append(float(factor_peek) if applied else -1.0)
        feed_in, _ = get_model_feed_in(batch, 'cpu')
        n_rays = feed_in['rays_o'].shape[1]
        rec['batch_sums'].append(U.batch_summary({k: feed_in[k][0].numpy() for k in U.BATCH_KEYS}))
        if epoch in (0, U.PRECROP_MAX_EPOCH) and 'batch{}_rays_o'.format(epoch) not in out and seed == U.SEEDS[0]:
            for k in U.BATCH_KEYS + ('rays_r',):
                out['batch{}_{}'.format(epoch, k)] = feed_in[k][0].numpy().copy()
        # ---- the step
        output = model(feed_in, get_progress=False, cur_epoch=epoch, total_epoch=int(cfgs.progress.epoch))
        loss = loss_factory(feed_in, output)
        optimizer.zero_grad()
        loss['sum'].backward()
        optimizer.step()
        ema.ema_step()
        rec['n_rays'].append(n_rays)
        rec['loss'].append(float(loss['sum']))
        rec['n_valid'].append(_state['n_valid'])
        if verbose and (epoch < 12 or epoch % 20 == 0 or refreshed and epoch < 64):
            print('{}epoch {:3d} rays {:4d} samples {:6d} loss {:9.4f} refreshed {} occ {} near {} [{:.0f} s]'.format(
                tag, epoch, n_rays, _state['n_valid'], float(loss['sum']), int(refreshed), rec['popcount'][-1] if rec['popcount'] else n_cells,
                rec['n_near'][-1] if rec['n_near'] else 0, time.time() - t_start), flush=True)
        if (epoch + 1) in U.SUMMARY_STEPS:
            pre = '{}p{}.'.format(tag, epoch + 1)
            for n, p in model.named_parameters():
                if not p.requires_grad:
                    continue
                if n.endswith('embed_fn.embeddings'):
                    t = p.detach().numpy()
                    out[pre + 'table.level_sum'] = np.stack([t[offsets[l]:offsets[l + 1]].astype(np.float64).sum(0) for l in range(len(offsets) - 1)])
                    out[pre + 'table.level_abs'] = np.stack([np.abs(t[offsets[l]:offsets[l + 1]]).astype(np.float64).sum(0) for l in range(len(offsets) - 1)])
                else:
                    out[pre + n] = p.detach().numpy().copy()
        if (epoch + 1) in U.CHECKPOINTS:
            p, ws = evaluate()
            rec['psnr'].append(p), rec['white_share'].append(ws)
            rec['occupied'].append(float(vol.get_voxel_bitfield(flatten=True).float().mean()))
            if verbose:
                print('{}after {:3d} iterations: held-out PSNR {:.2f} dB, white share {:.3f}, occupied {:.4f}, rays {} [{:.0f} s]'.format(
                    tag, epoch + 1, p, ws, rec['occupied'][-1], pipe.get_info('n_rays'), time.time() - t_start), flush=True)
    for kk, vv in rec.items():
        out[tag + kk] = np.array(vv)
    out[tag + 'bitfields'] = np.stack(bitfields) if bitfields else np.zeros((0, n_cells // 8), np.uint8)
    out[tag + 'near'] = np.stack(near) if near else np.zeros((0, n_cells // 8), np.uint8)
    out[tag + 'final_n_rays'] = np.array(pipe.get_info('n_rays'))
    out[tag + 'sampler_state'] = np.array([_state['rng'].state, _state['rng'].inc], np.uint64)


def run_data_only(rgba_train, out, seed=0):
    """the reference's Pipeline alone, precrop.max_epoch = U.QUIRK_MAX_EPOCH: batches of U.QUIRK_EPOCHS iterations"""
    ov = list(OVERRIDES)
    ov[ov.index('--dataset.train.scheduler.precrop.max_epoch') + 1] = str(U.QUIRK_MAX_EPOCH)
    cfgs = load_configs(EXPR, ov)
    log = Log()
    pipe = Pipeline()
    pipe.set_n_rays(log, get_value_from_cfgs_field(cfgs, 'n_rays', 1024))
    pipe.setup_cfgs(get_value_from_cfgs_field(cfgs.dataset.train, 'scheduler', None))
    img, mask = (torch.from_numpy(a) for a in U.dataset_tensors(rgba_train))
    rays_o, rays_d, rays_r = view_rays(U.TRAIN_VIEWS[0], U.N_TRAIN)

    def set_train_dataset():
        return {'img': img.clone(), 'mask': mask.clone(), 'rays_o': rays_o.clone(), 'rays_d': rays_d.clone(), 'rays_r': rays_r.clone(), 'H': U.H, 'W': U.W}

    n_shuffle = [0]

    def process(data):
        k = n_shuffle[0]
        with Fed(perm=lambda n: U.shuffle_perm(seed, k, n)):
            data = pipe.process_train_data(log, data)
        n_shuffle[0] += 1
        return data

    data = process(set_train_dataset())
    n_rays, sums, shuffle_at, crop_state = [], [], [], []
    for epoch in range(U.QUIRK_EPOCHS):
        crop_shuffle, full_shuffle = pipe.check_crop_shuffle(epoch), pipe.check_full_shuffle()
        if crop_shuffle:
            data = process(set_train_dataset())
        elif full_shuffle:
            data = process(data)
        if crop_shuffle or full_shuffle:
            shuffle_at.append(epoch)
        with Fed(uni=lambda shape: U.bkg_draw(seed, epoch, shape[1])[None]):
            batch = pipe.get_train_batch(data, epoch, None)
        feed_in, _ = get_model_feed_in(batch, 'cpu')
        n_rays.append(feed_in['rays_o'].shape[1])
        sums.append(U.batch_summary({k: feed_in[k][0].numpy() for k in U.BATCH_KEYS}))
        crop_state.append(-1 if pipe.crop_max_epoch is None else int(pipe.crop_max_epoch))
    out['quirk_n_rays'], out['quirk_batch_sums'], out['quirk_shuffle_at'] = np.array(n_rays), np.array(sums), np.array(shuffle_at)
    out['quirk_crop_max_epoch'] = np.array(crop_state)
    out['quirk_total_samples'] = np.array(pipe.get_info('total_samples'))
    print('quirk leg: reshuffles at', shuffle_at, 'total samples at the end', pipe.get_info('total_samples'), 'crop_max_epoch', crop_state[0], '->', crop_state[-1])


def main():
    """no argument: every seed in this process; `part <seed>`: one seed -> /tmp/g27_part_<seed>.npz (run the seeds side by side);
    `merge`: the parts -> the fixture"""
    torch.set_num_threads(int(os.environ.get('G27_THREADS', '8')))
    rgba_train, rgba_test = render_images()
    out = {'rgba_train': rgba_train, 'rgba_test': rgba_test}
    Ks, Ms = U.cameras(U.TRAIN_VIEWS[0], U.N_TRAIN)
    Kt, Mt = U.cameras(U.TEST_VIEWS[0], U.N_TEST)
    out['K_train'], out['c2w_train'], out['K_test'], out['c2w_test'] = Ks, Ms, Kt, Mt
    if len(sys.argv) > 2 and sys.argv[1] == 'part':
        part = {}
        run(int(sys.argv[2]), rgba_train, rgba_test, part)
        np.savez_compressed('/tmp/g27_part_{}.npz'.format(int(sys.argv[2])), **part)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'merge':
        for seed in SEEDS:
            part = np.load('/tmp/g27_part_{}.npz'.format(seed))
            for k in part.files:
                if k not in out:
                    out[k] = part[k]
                elif not k.startswith('s{}_'.format(seed)):
                    assert np.array_equal(out[k], part[k]), k       # the seed-independent entries agree
    else:
        for seed in SEEDS:
            run(seed, rgba_train, rgba_test, out)
    run_data_only(rgba_train, out)
    if DRY:
        print('dry run: nothing written')
        return
    path = os.path.join(OUT, 'g27_psnr.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')
    for c, cp in enumerate(U.CHECKPOINTS):
        print('after', cp, 'iterations: PSNR', [round(float(out['s{}_psnr'.format(s)][c]), 2) for s in SEEDS],
              'white share', [round(float(out['s{}_white_share'.format(s)][c]), 3) for s in SEEDS])


if __name__ == '__main__':
    main()
