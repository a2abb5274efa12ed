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


def neus(fixture, seed):
    torch.manual_seed(seed)
    model = build_model(load_configs('/root/reference/configs/models/neus.yaml', list(OVERRIDES)), None)
    out = {'overrides': np.array(OVERRIDES)}
    reseed_big_weights(model, out, seed)
    g, inputs = rays(seed + 1, 1, 40, 3.0, 1.6)     # some rays miss the radius-1.5 sphere
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.detach().numpy()
    model.fg_model.set_ray_cfgs('perturb', False)
    model.fg_model.set_ray_cfgs('noise_std', 0.0)
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
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
    print('neus: hit rays', int((res['mask'] > 0).sum()), 'of 40, scale', out['train_scale'], 'loss', float(loss), 'eik', float(eik),
          'mask mean', float(res['mask'].mean()))
    save(fixture, out)


if __name__ == '__main__':
    nerf_like('nerf', 'g22_nerf_fullwidth', 2200)
    neus('g23_neus_fullwidth', 2300)
    nerf_like('hdrnerf', 'g24_hdrnerf_fullwidth', 2400, hdr=True)
