"""G19: surface rendering vectors from the reference (run in the build container only).

FullModel.surface_render of the reference on the models of G13 (NeuS, configs/models/neus.yaml reduced: sphere tracing and the
secant search on the sdf's zero level) and G9 (vanilla NeRF, configs/models/nerf.yaml reduced: the secant search on a density
level chosen so that part of the rays cross it), with the state_dicts those fixtures hold.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
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

from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def load(fixture, cfg):
    g = np.load(os.path.join(OUT, fixture))
    model = build_model(load_configs('/root/reference/configs/models/' + cfg, [str(v) for v in g['overrides']]), None)
    model.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('in_')}
    return model.eval(), inputs


def main():
    out = {}
    neus, inputs = load('g13_neus_model.npz', 'neus.yaml')
    for tag, kw in (('neus_st_', dict(method='sphere_tracing', n_iter=60, threshold=0.002)),
                    ('neus_sec_', dict(method='secant_root_finding', n_step=48, n_iter=12, threshold=0.002))):
        res = neus.surface_render({k: v.clone() for k, v in inputs.items()}, **kw)
        for k, v in res.items():
            out[tag + k] = v.detach().numpy()
        print(tag, {k: tuple(v.shape) for k, v in res.items()}, 'hits', float(res['mask'].sum()), 'of', res['mask'].numel())
    nerf, inputs = load('g9_nerf_model.npz', 'nerf.yaml')
    with torch.no_grad():   # pick a density level that some, not all, of these rays cross
        o, d = inputs['rays_o'].view(-1, 3), inputs['rays_d'].view(-1, 3)
        t = torch.linspace(2.0, 6.0, 64)[None, :, None]
        sig = nerf.forward_pts((o[:, None] + d[:, None] * t).view(-1, 3)).view(o.shape[0], -1)
        level = float(torch.quantile(sig.max(dim=1)[0], 0.4))
    out['nerf_level'] = np.float32(level)
    res = nerf.surface_render({k: v.clone() for k, v in inputs.items()}, method='secant_root_finding', n_step=48, n_iter=12,
                              threshold=0.002, level=level, grad_dir='descent')
    for k, v in res.items():
        out['nerf_sec_' + k] = v.detach().numpy()
    print('nerf level', level, {k: tuple(v.shape) for k, v in res.items()}, 'hits', float(res['mask'].sum()), 'of', res['mask'].numel())
    path = os.path.join(OUT, 'g19_surface_render.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')


if __name__ == '__main__':
    main()
