"""G14: end-to-end HDR-NeRF vectors from the reference (run in the build container only, like make_golden_models.py).

configs/models/hdrnerf.yaml with the widths reduced: FullModel forward with per-ray exposure times in inference_only mode
(rgb / hdr / depth / mask) and in train mode with perturb / noise off (coarse + fine keys incl. hdr and unit_exp), plus the
gradients of the LDR MSE + a unit-exposure term.  state_dict exported alongside.
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
OVERRIDES = ['--model.geometry.W', '64', '--model.geometry.W_feat', '64', '--model.radiance.W', '32', '--model.radiance.W_feat_in', '64',
             '--model.exp_mlps.W', '16', '--model.chunk_pts', '4096', '--model.rays.n_sample', '32', '--model.rays.n_importance', '32']


def main():
    torch.manual_seed(1414)
    cfgs = load_configs('/root/reference/configs/models/hdrnerf.yaml', list(OVERRIDES))
    model = build_model(cfgs, None)
    g = torch.Generator().manual_seed(1415)
    B, N = 2, 64
    o = torch.randn(B, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 4.0
    d = -o + (torch.rand(B, N, 3, generator=g) - 0.5) * 1.5
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(B, N, 1), 'img': torch.rand(B, N, 3, generator=g),
              'bkg_color': torch.rand(B, N, 3, generator=g), 'exp_time': torch.rand(B, N, 1, generator=g) * 4.0 + 0.1}
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith('geo_net.layers.8.weight'):
                p[:1] += 0.6          # make the density matter
    out = {}
    with torch.no_grad():
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.numpy()
    model.fg_model.set_ray_cfgs('perturb', False)
    model.fg_model.set_ray_cfgs('noise_std', 0.0)
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    unit = sum(((res['unit_exp_' + s] - 0.5) ** 2).mean() for s in ('coarse', 'fine'))
    loss = ((res['rgb_fine'] - inputs['img']) ** 2).mean() + ((res['rgb_coarse'] - inputs['img']) ** 2).mean() + 0.5 * unit
    loss.backward()
    out['train_loss'] = loss.detach().numpy()
    for k, v in res.items():
        if torch.is_tensor(v):
            out['train_' + k] = v.detach().numpy()
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    for k, v in model.state_dict().items():
        out['sd.' + k] = v.numpy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['grad.' + k] = p.grad.numpy()
    out['overrides'] = np.array(OVERRIDES)
    path = os.path.join(OUT, 'g14_hdrnerf_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB', sorted(k for k in out if not k.startswith(('sd.', 'grad.'))))
    print({k: tuple(v.shape) for k, v in res.items() if torch.is_tensor(v)})


if __name__ == '__main__':
    main()
