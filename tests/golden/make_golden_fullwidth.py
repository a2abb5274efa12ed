"""G22 / G23 / G24: the reference's FullModel at the FULL widths of BASELINE configs 1 / 3 / 5 (run in the build container only).

configs/models/nerf.yaml (8x256 skip@4 + 128, coarse + fine, 64+128 samples), neus.yaml (8x256 softplus-100 geometric init, weight
norm, skip-reduce, 4x256 radiance, 64+64 samples in 4 up-sampling rounds) and hdrnerf.yaml (nerf.yaml + three 1->128->1 tone
mappers) with NO width overrides, few rays.  The earlier model fixtures G9 / G13 / G14 shrink the nets to W = 64; these do not.

Matrices with more than 4096 elements are not stored: they are regenerated from a seed and the per-column mean / std of the
reference's own initialisation (tests/seeded_weights.py), loaded into the reference model here and into the mirror in the tests;
their gradients are stored as summaries.  Small tensors (biases, weight-norm g, variance, tone-mapper layers) are stored verbatim.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, os.path.dirname(HERE))
sys.path.insert(2, HERE)
_r = types.ModuleType('pytorch3d.transforms.rotation_conversions')
for _n in ['axis_angle_to_matrix', 'matrix_to_axis_angle', 'matrix_to_rotation_6d', 'rotation_6d_to_matrix']:
    setattr(_r, _n, lambda *a, **k: None)
sys.modules['pytorch3d'] = types.ModuleType('pytorch3d')
sys.modules['pytorch3d.transforms'] = types.ModuleType('pytorch3d.transforms')
sys.modules['pytorch3d.transforms.rotation_conversions'] = _r
import warnings  # noqa: E402

warnings.filterwarnings('ignore')
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seeded_weights as SW  # noqa: E402
from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402

OVERRIDES = ['--model.chunk_pts', '8192']


def reseed_big_weights(model, out, seed):
    """replace every big matrix by its seeded twin with the same column statistics; keep weight-norm g consistent (g = |v| rows, the
    state nn.utils.weight_norm starts from)"""
    out['weight_seed'] = np.array(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 2 and p.numel() > SW.BIG:
                w = p.numpy()
                mean, std = w.mean(0).astype(np.float32), w.std(0).astype(np.float32)
                out['seeded_mean.' + name], out['seeded_std.' + name] = mean, std
                out['seeded_shape.' + name] = np.array(w.shape)
                p.copy_(torch.from_numpy(SW.make_weight(name, w.shape, seed, mean, std)))
        for name, p in model.named_parameters():
            if name.endswith('weight_g'):
                v = dict(model.named_parameters())[name[:-1] + 'v']
                p.copy_(v.norm(dim=1, keepdim=True))
    for k, v in model.state_dict().items():
        if ('seeded_mean.' + k) not in out:
            out['sd.' + k] = v.numpy().copy()


def store_grads(model, out, tag='grad.'):
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if ('seeded_mean.' + k) in out:
            for kk, vv in SW.grad_summary(p.grad.numpy()).items():
                out['gsum.' + k + '.' + kk] = vv
        else:
            out[tag + k] = p.grad.numpy().copy()


def rays(seed, B, N, radius, spread):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(B, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * radius
    d = -o + (torch.rand(B, N, 3, generator=g) - 0.5) * spread
    d = d / d.norm(dim=-1, keepdim=True)
    return g, {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(B, N, 1), 'img': torch.rand(B, N, 3, generator=g),
               'bkg_color': torch.rand(B, N, 3, generator=g)}


def save(name, out):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')


def nerf_like(cfg_name, fixture, seed, hdr=False):
    torch.manual_seed(seed)
    model = build_model(load_configs('/root/reference/configs/models/{}.yaml'.format(cfg_name), list(OVERRIDES)), None)
    out = {'overrides': np.array(OVERRIDES)}
    reseed_big_weights(model, out, seed)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith('geo_net.layers.8.weight'):
                p[:1] += 0.35          # make the density matter (default init: sigma ~ 0); the edited row of the seeded matrix is stored
                out['rowpatch.' + n_] = p[:1].numpy().copy()
    g, inputs = rays(seed + 1, 1, 40, 4.0, 1.5)
    if hdr:
        inputs['exp_time'] = torch.rand(1, 40, 1, generator=g) * 4.0 + 0.1
    with torch.no_grad():
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.numpy()
    model.fg_model.set_ray_cfgs('perturb', False)
    model.fg_model.set_ray_cfgs('noise_std', 0.0)
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    loss = ((res['rgb_fine'] - inputs['img']) ** 2).mean() + ((res['rgb_coarse'] - inputs['img']) ** 2).mean()
    if hdr:
        loss = loss + 0.5 * sum(((res['unit_exp_' + s] - 0.5) ** 2).mean() for s in ('coarse', 'fine'))
    loss.backward()
    out['train_loss'] = loss.detach().numpy()
    for k, v in res.items():
        if torch.is_tensor(v):
            out['train_' + k] = v.detach().numpy()
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    store_grads(model, out)
    print(cfg_name, 'loss', float(loss), 'mask mean', float(res['mask_fine'].mean()), {k: tuple(v.shape) for k, v in res.items() if torch.is_tensor(v)})
    save(fixture, out)


def neus(fixture, seed, pool=400, keep=40, margin=2e-6, pos_noise=5e-6):
    """G23.  Inference pass (deterministic lattice) + TRAINING pass with `perturb: True` as configs/models/neus.yaml has it, the
    reference's torch.rand draws on tape (tie_probe.RandTape) so the mirror can be fed the same uniforms.  Why not perturb off as in the
    first version of this fixture: the deterministic lattice contains u = 1.0, compared with a float cumsum that is 1.0 to an ulp - a
    coin flip on the last sample of the first up-sampling round of every ray that hits (tie_probe.py) - and the Eikonal gradient of the
    matrices fed by the 2^9-frequency embedding feels one moved sample at the 1e-2 level.  With recorded uniforms every inverse-CDF
    decision of the stored rays has a margin >= `margin` in cdf units (they are picked from a pool by that criterion; indices and
    margins are stored), so the gradients can be held at 1e-3 without exceptions."""
    import arcnerf.models.neus_model as NM
    import arcnerf.render.ray_helper as RH
    from tie_probe import ProbeU, RandTape, float64_gradients, inference_flip_sensitivity, inference_lattice_margin, store_fp32_error
    torch.manual_seed(seed)
    model = build_model(load_configs('/root/reference/configs/models/neus.yaml', list(OVERRIDES)), None)
    out = {'overrides': np.array(OVERRIDES)}
    reseed_big_weights(model, out, seed)
    g, pool_in = rays(seed + 1, 1, pool, 3.0, 1.6)
    n_miss = 4                                         # the last rays of the pool pass the radius-1.5 sphere by: invalid-ray defaults
    tang = torch.cross(pool_in['rays_o'][0, -n_miss:], torch.tensor([0.0, 0.0, 1.0]).expand(n_miss, 3), dim=-1)
    pool_in['rays_d'][0, -n_miss:] = tang / tang.norm(dim=-1, keepdim=True)
    model.fg_model.set_ray_cfgs('noise_std', 0.0)
    assert model.fg_model.get_ray_cfgs('perturb') is True
    flat = {k: v.view(-1, v.shape[-1]) for k, v in pool_in.items()}
    hit = model.fg_model.obj_bound.get_near_far_from_rays(flat)[2]
    hit = torch.ones(pool, dtype=torch.bool) if hit is None else hit.view(-1)
    tape = RandTape(seed + 2)
    with tape.record(), ProbeU(tape, RH, NM) as probe:
        model({k: v.clone() for k, v in pool_in.items()}, inference_only=False, cur_epoch=20000)
    pool_draws = [d.clone() for d in tape.draws]
    m_hit = probe.per_ray()[1]
    n_hit = int(hit.sum())
    assert m_hit.shape[0] == n_hit and all(d.shape[0] in (n_hit, pool) for d in pool_draws)   # coarse depths: all rays; up-sampling: the hits
    m_pool = np.full(pool, 1.0)
    m_pool[hit.numpy()] = m_hit
    assert int((~hit[-n_miss:]).sum()) == n_miss
    # inference runs on the deterministic lattice, whose u = 1.0 is a coin flip against cdf[-1] (tie_probe.py): keep only rays whose
    # inference outputs are the same either way
    flip = inference_flip_sensitivity(model, pool_in, RH)
    lat = np.full(pool, 1.0)
    noise = np.zeros(pool)
    lat[hit.numpy()], noise[hit.numpy()] = inference_lattice_margin(model, pool_in, RH, NM)
    m_pool = np.minimum(m_pool, lat)      # one margin for both passes: taped uniforms (training) and lattice points (inference)
    noise[hit.numpy()] = np.maximum(noise[hit.numpy()], probe.per_ray_noise())   # worst-conditioned sample of either pass (tie_probe.position_noise)
    ok = np.nonzero((m_pool >= margin) & hit.numpy() & (flip < 1e-5) & (noise <= pos_noise))[0]
    print('rays with a sample in an ill-conditioned bin (position noise >', pos_noise, '):', int((noise > pos_noise).sum()), 'of', pool)
    sel = np.sort(np.concatenate([ok[:keep - n_miss], np.arange(pool - n_miss, pool)]))
    assert len(sel) == keep, (len(ok), keep)
    out['flip_sensitivity'], out['position_noise'] = flip[sel], noise[sel]
    print('neus: rays whose inference outputs depend on the u = 1 decision:', int((flip >= 1e-5).sum()), 'of', pool)
    rank = np.cumsum(hit.numpy()) - 1                  # pool ray -> row of the draws
    rows = rank[sel][hit.numpy()[sel]]
    inputs = {k: v[:, sel].contiguous() for k, v in pool_in.items()}
    out['pool_size'], out['pool_sel'], out['tie_margin'] = np.array(pool), sel, m_pool[sel]
    print('neus: pool', pool, 'hit', int(hit.sum()), 'rays with margin <', margin, ':', int((m_pool < margin).sum()), 'kept', keep,
          'of which hit', int(hit.numpy()[sel].sum()), 'min margin kept', m_pool[sel].min())
    # inference (deterministic lattice): outputs only
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.detach().numpy()
    # training pass on the kept rays with their rows of the taped draws
    draws = [d[sel] if d.shape[0] == pool else d[rows] for d in pool_draws]
    seen_z, real_mid = [], model.fg_model.handle_mid_pts
    model.fg_model.handle_mid_pts = lambda z, mk: (seen_z.append(z.detach().clone()), real_mid(z, mk))[1]
    with tape.replay(draws), ProbeU(tape, RH, NM) as probe:
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    model.fg_model.handle_mid_pts = real_mid
    out['train_zvals_upsampled'] = seen_z[0].numpy()      # (hit rays, 128): the depths after the four up-sampling rounds
    assert probe.per_ray()[1].min() >= margin, probe.per_ray()[1].min()
    for i, d in enumerate(draws):
        out['draw_{:02d}'.format(i)] = d.numpy()
    eik = ((res['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((res['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    loss.backward()
    out['train_loss'], out['train_eikonal'] = loss.detach().numpy(), eik.detach().numpy()
    for k, v in res.items():
        if torch.is_tensor(v):
            out['train_' + k] = v.detach().numpy()
    prm = res['params'][0] if isinstance(res['params'], list) else res['params']
    out['train_scale'] = np.float32(prm['scale'])
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    store_grads(model, out)
    g32 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    g64, o64, l64 = float64_gradients(model, inputs, draws, lambda r, i: ((r['rgb'] - i['img']) ** 2).mean() + 0.1 * ((r['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean(), cur_epoch=20000)
    worst = store_fp32_error(out, '', g32, g64, o64, res)
    print('neus: the reference fp32 gradient against its float64 evaluation, relative to the max: worst',
          sorted(((round(v, 5), k) for k, v in worst.items()), reverse=True)[:4], 'loss', float(loss), l64)
    print('neus: hit rays', int((res['mask'] > 0).sum()), 'of', keep, 'scale', out['train_scale'], 'loss', float(loss), 'eik', float(eik),
          'mask mean', float(res['mask'].mean()), 'draws', [tuple(d.shape) for d in draws])
    save(fixture, out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g22', 'g23', 'g24']
    if 'g22' in which:
        nerf_like('nerf', 'g22_nerf_fullwidth', 2200)
    if 'g23' in which:
        neus('g23_neus_fullwidth', 2300)
    if 'g24' in which:
        nerf_like('hdrnerf', 'g24_hdrnerf_fullwidth', 2400, hdr=True)
