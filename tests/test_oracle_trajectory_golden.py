"""Pins oracle/ngp_trainer.py (the reference's instant-ngp training LOOP restated on the CPU oracle: VolumeBound.optimize, the dynamic
batch size, oracle_step, torch.optim.Adam's update, EMA.ema_step) to golden G26 - 2 x 20 steps of the reference's own loop
(tests/golden/make_golden_trajectory.py).  The GPU tests (tests/test_gpu_trajectory.py) hold the product to the same fixture, and use this
oracle for the fused-net semantics the reference cannot run on a CPU."""
import numpy as np
import pytest

import g26_utils as U


def make_cfg(**kw):
    from arcnerf_amd.pipeline import NgpConfig
    base = dict(geo_fused_semantics=False, has_bias=False, W_feat=15, add_inf_z=False, noise_std=0.0, white_bkg=True, n_grid=U.N_GRID,
                n_sample=U.N_SAMPLE, epoch_optim=U.EPOCH_OPTIM, epoch_optim_warmup=U.EPOCH_WARMUP)
    base.update(kw)
    return NgpConfig(**base)


def flat_params(g, leg, fld):
    """the leg's start state in the flat layout of an NgpField: seeded table + the stored MLP weights"""
    flat = np.zeros(fld.n_params, np.float32)
    off, n = fld._seg['table']
    flat[off:off + n] = U.table(g, leg, fld.offsets[-1]).reshape(-1)
    for name, fmt, k in (('geo', 'fg_model.coarse_geo_net.layers.{}.weight', 2), ('rad', 'fg_model.coarse_radiance_net.layers.{}.weight', 3)):
        off, n = fld._seg[name + '_w']
        w = np.concatenate([g['{}_sd.{}'.format(leg, fmt.format(i))].reshape(-1) for i in range(k)])
        assert w.shape[0] == n
        flat[off:off + n] = w
    return flat


def nets_of(fld, flat):
    out = {}
    for name, fmt, dims in (('geo', 'fg_model.coarse_geo_net.layers.{}.weight', fld.geo_dims), ('rad', 'fg_model.coarse_radiance_net.layers.{}.weight', fld.rad_dims)):
        off, _ = fld._seg[name + '_w']
        for i in range(len(dims) - 1):
            k = dims[i] * dims[i + 1]
            out[fmt.format(i)] = flat[off:off + k].reshape(dims[i + 1], dims[i])
            off += k
    return out


@pytest.mark.parametrize('leg', ['a', 'b'])
def test_oracle_loop_matches_reference_loop(oracle, leg):
    from arcnerf_amd.pipeline import NgpField
    from oracle.ngp_trainer import OracleNgpTrainer
    g = U.golden()
    U.check_inputs(g, leg)
    cfg = make_cfg()
    assert [cfg.lr, cfg.eps, cfg.weight_decay, cfg.ema_decay] == [float(v) for v in g['optim']]
    assert [cfg.huber_delta, cfg.loss_weight] == [float(v) for v in g['loss_cfg']]
    fld = NgpField(cfg, device='cpu', seed=0)
    assert fld.offsets == [int(v) for v in g['offsets']]
    epochs = U.LEGS[leg]['epochs']
    tr = OracleNgpTrainer(oracle, fld, cfg, flat_params(g, leg, fld), 1 << U.LOG_MAX_ALLOWANCE, U.N_RAYS0, U.UPDATE_EPOCH, U.N_RAYS_MAX,
                          start_epoch=epochs[0]).start_ema()
    bars = U.loss_bars(g, leg)
    n_ref, flips = 0, 0
    for k, epoch in enumerate(epochs):
        perm, uni = U.refresh_draws(epoch, cfg.n_grid ** 3)
        if tr.optimize(epoch, perm, uni):
            assert int(g[leg + '_refreshed'][k]) == 1 and tr.last['n_refresh_pts'] == int(g[leg + '_n_refresh_pts'][n_ref])
            assert abs(tr.last['thres'] - float(g[leg + '_thres'][n_ref])) <= (1e-5 if flips == 0 else 1e-2) * float(g[leg + '_thres'][n_ref])
            flips += U.check_bitfield(g, leg, n_ref, tr.bitfield, flips)
            n_ref += 1
        else:
            assert int(g[leg + '_refreshed'][k]) == 0
        n_rays = tr.update_n_rays(epoch)
        assert n_rays == int(g[leg + '_n_rays'][k]), (leg, k, n_rays)
        if float(g[leg + '_dyn_factor'][k]) > 0:
            assert abs(tr.last['dyn_factor'] - float(g[leg + '_dyn_factor'][k])) <= 1e-3 * float(g[leg + '_dyn_factor'][k])
        inp = U.step_inputs(epoch, n_rays)
        res = tr.step(inp['rays_o'], inp['rays_d'], inp['bkg_color'], inp['img'])
        if flips == 0:
            assert res['n_samples'] == int(g[leg + '_n_valid'][k]), (leg, k)
        else:
            assert abs(res['n_samples'] - int(g[leg + '_n_valid'][k])) <= 0.01 * int(g[leg + '_n_valid'][k])
        rel = abs(res['loss'] - float(g[leg + '_loss'][k])) / float(g[leg + '_loss'][k])
        assert rel <= bars[k], (leg, 'step', k + 1, 'loss', res['loss'], float(g[leg + '_loss'][k]), rel, bars[k])
        if (k + 1) in U.SUMMARY_STEPS:
            rep = U.param_report(g, leg, k + 1, tr.views()['table'], nets_of(fld, tr.p))
            U.check_params(rep, bars[k] <= 1e-4 and flips == 0, (leg, k + 1))
    assert n_ref == len(g[leg + '_bitfields'])
