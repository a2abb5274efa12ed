"""Golden G27 on the CPU (tests/golden/make_golden_psnr.py: the reference's training loop WITH ITS OWN Pipeline, 600 iterations x 4 seeds):

  1. oracle/batch.py (the batch fetch restated with numpy) against the reference's batches: two in full, all 2400 by checksums;
  2. the HOST logic of the product's trainer.Pipeline - crop window, composed shuffles, pass bookkeeping, the short last batch of a pass, the
     end of the crop, the dynamic batch size - with its one launch replaced by that restatement: the same 2400 batches;
  3. oracle/ngp_trainer.py + oracle/batch.py against the first iterations of the reference's loop (losses, sample counts, bitfields).
The GPU tests (tests/test_gpu_psnr.py) hold the kernels, the module API and the fused step to the same fixture."""
import numpy as np
import pytest
import torch

import g27_utils as U


@pytest.fixture(scope='module')
def g():
    return U.golden()


class OracleFetch:
    """stands in for arcnerf_amd.ops.functional.fetch_train_batch (torch CPU tensors in and out), computing with oracle/batch.py"""

    def __init__(self):
        self.rays = {}

    def __call__(self, ids, n_img, H, W, window=None, rgba=None, img=None, mask=None, intrinsic=None, c2w=None, center_pixel=True,
                 normalize_rays_d=True, bkg_rand=None, bkg_const=None, want_rays_r=True, want_src=False, bad_ids=None):
        from oracle import batch as OB
        npy = lambda t: None if t is None else t.numpy()
        rays = None
        if intrinsic is not None:
            key = (intrinsic.data_ptr(), c2w.data_ptr(), H, W)
            if key not in self.rays:
                self.rays[key] = OB.dataset_rays(H, W, npy(intrinsic), npy(c2w), center_pixel, normalize_rays_d)
            rays = self.rays[key]
        out = OB.fetch_train_batch(npy(ids), n_img, H, W, window, rgba=npy(rgba), img=npy(img), mask=npy(mask), bkg_rand=npy(bkg_rand),
                                   bkg_const=bkg_const, rays=rays)
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}


def dataset(g, form='rgba'):
    rgba = g['rgba_train']
    n = rgba.shape[0]
    d = {'H': U.H, 'W': U.W, 'intrinsic': torch.from_numpy(g['K_train']), 'c2w': torch.from_numpy(g['c2w_train'])}
    if form == 'rgba':
        d['rgba'] = torch.from_numpy(rgba.reshape(n, -1, 4).copy())
    else:
        img, mask = U.dataset_tensors(rgba)
        d['img'], d['mask'] = torch.from_numpy(img), torch.from_numpy(mask)
    return d


def scheduler_cfg(g):
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    ratio, max_epoch, update_epoch, max_bs = [float(v) for v in g['scheduler']]
    return dict_to_obj({'precrop': {'ratio': ratio, 'max_epoch': int(max_epoch)}, 'bkg_color': {'color': 'random'},
                        'dynamic_batch_size': {'update_epoch': int(update_epoch), 'max_batch_size': int(max_bs)}})


class FactorTape:
    """get_dynamicbs_factor() of the run in the fixture (the factor depends on the model's samples: not this test's subject)"""

    def __init__(self, g, seed):
        self.f = g['s{}_dyn_factor'.format(seed)]
        self.epoch = 0

    def get_dynamicbs_factor(self):
        assert self.f[self.epoch] > 0, self.epoch
        return float(self.f[self.epoch])


def check_sums(got, want, where):
    """rays to the distance between two correct float32 evaluations of get_rays (1e-6 per component), everything else exactly"""
    n = got['rays_o'].shape[0]
    s = U.batch_summary(got)
    for i, k in enumerate(U.BATCH_KEYS):
        tol = (2e-6 * n * 3, 2e-6 * n * 3 * n) if k.startswith('rays') else (1e-9 * max(1.0, abs(want[i, 0])), 1e-9 * max(1.0, abs(want[i, 1])))
        assert abs(s[i, 0] - want[i, 0]) <= tol[0] and abs(s[i, 1] - want[i, 1]) <= tol[1], (where, k, s[i], want[i])


def test_oracle_batch_fetch_matches_the_reference_pipeline(oracle, g):
    from oracle import batch as OB
    rgba = g['rgba_train']
    n_img = rgba.shape[0]
    rays = OB.dataset_rays(U.H, U.W, g['K_train'], g['c2w_train'])
    dh = int((1 - U.PRECROP_RATIO) * U.H / 2.0)
    crop = (dh, dh, U.H - 2 * dh, U.W - 2 * dh)
    seed = U.SEEDS[0]
    for epoch, window, k_shuffle in ((0, crop, 0), (U.PRECROP_MAX_EPOCH, None, 1)):
        total = n_img * (window[2] * window[3] if window else U.H * U.W)
        ids = U.shuffle_perm(seed, k_shuffle, total)[:U.N_RAYS0]
        out = OB.fetch_train_batch(ids, n_img, U.H, U.W, window, rgba=rgba, bkg_rand=U.bkg_draw(seed, epoch, U.N_RAYS0), rays=rays)
        for k in ('img', 'mask', 'bkg_color'):
            assert np.array_equal(out[k], g['batch{}_{}'.format(epoch, k)]), (epoch, k)
        for k in ('rays_o', 'rays_d'):
            assert np.abs(out[k] - g['batch{}_{}'.format(epoch, k)]).max() <= 1e-6, (epoch, k)
        assert np.allclose(out['rays_r'], g['batch{}_rays_r'.format(epoch)], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('form', ['rgba', 'float'])
def test_pipeline_host_logic_hands_out_the_reference_batches(oracle, g, form, monkeypatch):
    from arcnerf_amd import trainer as T
    from arcnerf_amd.ops import functional as F
    monkeypatch.setattr(F, 'fetch_train_batch', OracleFetch())
    monkeypatch.setattr(F, '_req', lambda *a: None)
    data = dataset(g, form)
    for seed in (U.SEEDS if form == 'rgba' else U.SEEDS[:1]):
        tag = 's{}_'.format(seed)
        p = T.Pipeline(tape=U.Tape(seed))
        p.set_n_rays(None, U.N_RAYS0)
        p.setup_cfgs(scheduler_cfg(g))
        batches = T.TrainBatches(p, lambda: data)
        model = FactorTape(g, seed)
        shuffles = [(-1, p.get_info('total_samples'))]
        want_n, sums = g[tag + 'n_rays'], g[tag + 'batch_sums']
        for epoch in range(len(want_n)):
            model.epoch = epoch
            k0 = p._n_shuffle
            n_rays = p.fetch_step_update_dynamic_bs(epoch, model)
            feed_in = batches(n_rays, epoch)
            if p._n_shuffle != k0:
                shuffles.append((epoch, p.get_info('total_samples')))
            assert feed_in['rays_o'].shape == (1, int(want_n[epoch]), 3), (seed, epoch, feed_in['rays_o'].shape, want_n[epoch])
            assert set(feed_in) == {'img', 'mask', 'rays_o', 'rays_d', 'rays_r', 'bkg_color'}
            check_sums({k: feed_in[k][0].numpy() for k in U.BATCH_KEYS}, sums[epoch], (seed, epoch))
        assert [s[0] for s in shuffles] == g[tag + 'shuffle_at'].tolist() and [s[1] for s in shuffles] == g[tag + 'shuffle_total'].tolist()
        assert p.get_info('n_rays') == int(g[tag + 'final_n_rays'])
        # the run saw the whole state machine: the crop, its end, a short last batch, a reshuffle of a finished pass, a changed batch size
        assert len(shuffles) >= 3 and shuffles[1][0] == U.PRECROP_MAX_EPOCH and len(set(want_n.tolist())) >= 3


def test_pipeline_follows_the_reference_when_a_pass_ends_before_the_crop(oracle, g, monkeypatch):
    """the fixture's data-only leg (the reference's Pipeline with precrop.max_epoch 100 > 59 batches of cropped rays): the reshuffle at
    iteration 59 clears crop_max_epoch and iteration 100 still draws from the centre windows - every batch of the 130 iterations"""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.ops import functional as F
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    monkeypatch.setattr(F, 'fetch_train_batch', OracleFetch())
    monkeypatch.setattr(F, '_req', lambda *a: None)
    p = T.Pipeline(tape=U.Tape(0))
    p.set_n_rays(None, U.N_RAYS0)
    p.setup_cfgs(dict_to_obj({'precrop': {'ratio': U.PRECROP_RATIO, 'max_epoch': U.QUIRK_MAX_EPOCH}, 'bkg_color': {'color': 'random'},
                              'dynamic_batch_size': {'update_epoch': U.UPDATE_EPOCH, 'max_batch_size': U.N_RAYS_MAX}}))
    batches = T.TrainBatches(p, lambda: dataset(g))
    shuffles = []
    for epoch in range(U.QUIRK_EPOCHS):
        k0 = p._n_shuffle
        feed_in = batches(p.fetch_step_update_dynamic_bs(epoch, None), epoch)
        if p._n_shuffle != k0:
            shuffles.append(epoch)
        assert feed_in['rays_o'].shape[1] == int(g['quirk_n_rays'][epoch])
        check_sums({k: feed_in[k][0].numpy() for k in U.BATCH_KEYS}, g['quirk_batch_sums'][epoch], ('quirk', epoch))
        assert (-1 if p.crop_max_epoch is None else p.crop_max_epoch) == int(g['quirk_crop_max_epoch'][epoch])
    assert shuffles == g['quirk_shuffle_at'].tolist() == [59, 118] and p.get_info('total_samples') == int(g['quirk_total_samples']) == U.N_TRAIN * 2500


def test_the_crop_never_ends_when_a_pass_finishes_first(oracle, g, monkeypatch):
    """the reference's state machine (pipeline.py:95-118): the second process_train_data call clears crop_max_epoch - if that call is the
    reshuffle of a finished pass over the cropped rays, check_crop_shuffle never fires again"""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.ops import functional as F
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    monkeypatch.setattr(F, 'fetch_train_batch', OracleFetch())
    monkeypatch.setattr(F, '_req', lambda *a: None)
    p = T.Pipeline(tape=U.Tape(0))
    p.set_n_rays(None, 4096)
    p.setup_cfgs(dict_to_obj({'precrop': {'ratio': 0.5, 'max_epoch': 50}, 'bkg_color': {'color': [1.0, 0.5, 0.0]}}))
    batches = T.TrainBatches(p, lambda: dataset(g))
    assert p.crop_max_epoch == 50 and p.get_info('total_samples') == 24 * 2500
    seen = []
    for epoch in range(60):
        f = batches(4096, epoch)
        seen.append(f['rays_o'].shape[1])
        assert torch.equal(f['bkg_color'][0], torch.tensor([1.0, 0.5, 0.0]).expand(seen[-1], 3))
    assert seen[:15] == [4096] * 14 + [24 * 2500 - 14 * 4096] and p.crop_max_epoch is None and p.get_info('total_samples') == 24 * 2500


def test_oracle_loop_follows_the_reference_loop_with_its_pipeline(oracle, g):
    """oracle/ngp_trainer.py on the batches of oracle/batch.py against the first 24 iterations of every seed's reference run"""
    from arcnerf_amd.pipeline import NgpConfig, NgpField
    from oracle import batch as OB
    from oracle.ngp_trainer import OracleNgpTrainer
    cfg = NgpConfig(geo_fused_semantics=False, has_bias=False, W_feat=15, add_inf_z=False, noise_std=0.0, white_bkg=True, n_grid=U.N_GRID,
                    n_sample=U.N_SAMPLE, epoch_optim=U.EPOCH_OPTIM, epoch_optim_warmup=U.EPOCH_WARMUP)
    assert [cfg.lr, cfg.eps, cfg.weight_decay, cfg.ema_decay] == [float(v) for v in g['optim']]
    rgba = g['rgba_train']
    n_img = rgba.shape[0]
    rays = OB.dataset_rays(U.H, U.W, g['K_train'], g['c2w_train'])
    dh = int((1 - U.PRECROP_RATIO) * U.H / 2.0)
    crop = (dh, dh, U.H - 2 * dh, U.W - 2 * dh)
    for seed in U.SEEDS[:2]:
        tag = 's{}_'.format(seed)
        fld = NgpField(cfg, device='cpu', seed=0)
        flat = start_state(g, seed, fld)
        tr = OracleNgpTrainer(oracle, fld, cfg, flat, 1 << U.LOG_MAX_ALLOWANCE, U.N_RAYS0, U.UPDATE_EPOCH, U.N_RAYS_MAX).start_ema()
        perm = U.shuffle_perm(seed, 0, n_img * crop[2] * crop[3])
        n_ref = flips = 0
        for epoch in range(24):
            pm, uni = U.refresh_draws(seed, epoch, cfg.n_grid ** 3)
            if tr.optimize(epoch, pm, uni):
                assert int(g[tag + 'refreshed'][epoch]) == 1
                ref = np.unpackbits(g[tag + 'bitfields'][n_ref], bitorder='little').astype(bool)
                near = np.unpackbits(g[tag + 'near'][n_ref], bitorder='little').astype(bool)
                diff = tr.bitfield != ref
                # (8 Adam steps at lr 1e-1, eps 1e-15 turn the last-ulp difference of the rays into a different SIGN of the rounding-noise
                # gradient entries, i.e. into rows that sit 2 lr apart: a few dozen cells of 32768 decide differently, near the threshold or not)
                assert int(diff.sum()) <= 0.005 * diff.size, (seed, epoch, int(diff.sum()), int((diff & ~near).sum()))
                flips += int(diff.sum())
                n_ref += 1
            b = OB.fetch_train_batch(perm[epoch * U.N_RAYS0:(epoch + 1) * U.N_RAYS0], n_img, U.H, U.W, crop, rgba=rgba,
                                     bkg_rand=U.bkg_draw(seed, epoch, U.N_RAYS0), rays=rays)
            res = tr.step(b['rays_o'], b['rays_d'], b['bkg_color'], b['img'])
            want_n, want_l = int(g[tag + 'n_valid'][epoch]), float(g[tag + 'loss'][epoch])
            if flips == 0:      # (the rays are the oracle's get_rays: 1e-6 from the reference's, a sample on a cell face may fall either way)
                assert abs(res['n_samples'] - want_n) <= max(4, 2e-4 * want_n), (seed, epoch, res['n_samples'], want_n)
                assert abs(res['loss'] - want_l) <= 2e-4 * want_l, (seed, epoch, res['loss'], want_l)
            else:
                assert abs(res['n_samples'] - want_n) <= 0.02 * want_n and abs(res['loss'] - want_l) <= 0.05 * want_l, (seed, epoch)
        assert n_ref == 2


def start_state(g, seed, fld):
    """the run's start state in the flat layout of an NgpField: the seeded table + the stored MLP weights"""
    flat = np.zeros(fld.n_params, np.float32)
    off, n = fld._seg['table']
    t = U.table_from_seed(fld.offsets[-1], 2, seed)
    assert abs(t.astype(np.float64).sum() - float(g['s{}_table_sum'.format(seed)])) < 1e-9 and np.array_equal(t[::100003], g['s{}_table_probe'.format(seed)])
    flat[off:off + n] = t.reshape(-1)
    for name, fmt, k in (('geo', 'fg_model.coarse_geo_net.layers.{}.weight', 2), ('rad', 'fg_model.coarse_radiance_net.layers.{}.weight', 3)):
        off, n = fld._seg[name + '_w']
        w = np.concatenate([g['s{}_sd.{}'.format(seed, fmt.format(i))].reshape(-1) for i in range(k)])
        assert w.shape[0] == n
        flat[off:off + n] = w
    return flat
