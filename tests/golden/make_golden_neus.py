"""G13: end-to-end NeuS vectors from the reference (run in the build container only, like make_golden_models.py).

configs/models/neus.yaml with the widths reduced (so the exported weights stay small): FullModel forward in inference_only
mode (deterministic: no perturbation, det inverse-CDF) on fixed rays -> rgb / depth / mask / normal; the train-mode keys with
perturb / noise off; and the gradients of rgb-MSE + 0.1 * Eikonal((|normal_pts| - 1)^2) - the Eikonal term makes the
parameters' gradient depend on the SECOND derivative of the geometry net (create_graph=True in forward_with_grad).
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
             '--model.chunk_pts', '4096', '--model.rays.n_sample', '32', '--model.rays.n_importance', '32']


def main():
    torch.manual_seed(1313)
    cfgs = load_configs('/root/reference/configs/models/neus.yaml', list(OVERRIDES))
    model = build_model(cfgs, None)
    g = torch.Generator().manual_seed(1314)
    B, N = 2, 80
    o = torch.randn(B, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 3.0
    d = -o + (torch.rand(B, N, 3, generator=g) - 0.5) * 1.6          # some rays miss the radius-1.5 sphere
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(B, N, 1), 'img': torch.rand(B, N, 3, generator=g),
              'bkg_color': torch.rand(B, N, 3, generator=g)}
    out = {}
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
    for k, v in model.state_dict().items():
        out['sd.' + k] = v.numpy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['grad.' + k] = p.grad.numpy()
    out['overrides'] = np.array(OVERRIDES)
    path = os.path.join(OUT, 'g13_neus_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB', sorted(k for k in out if not k.startswith(('sd.', 'grad.'))))
    print('hit rays', int((res['mask'] > 0).sum()), 'of', B * N, 'scale', out['train_scale'], 'loss', float(loss), 'eik', float(eik))


if __name__ == '__main__':
    main()
