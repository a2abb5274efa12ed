"""G18: NeuS on a hash grid (second-order path) from the reference, run in the build container only.

configs/neus_ngp_multivol.yaml of this repo (= the model block of the reference's capture_qqtiger_neusngp_multivol.yaml) with
what cannot run on CPU replaced: the occupancy-pruned volume (CUDA sampler) by the sphere bound of configs/models/neus.yaml, the
tcnn back-ends by the reference's torch back-ends, no background; the hash grid shrunk.  What it pins is the sdf net ON the hash
encoder with normals taken by autograd (create_graph) and the rgb + Eikonal loss reaching the table through them.  The edited
YAML travels inside the fixture so the test builds the very same model.
"""
import os
import sys
import tempfile
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
import yaml  # noqa: E402

from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


def edited_config():
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml')))
    m = cfg['model']
    del m['background']
    m['obj_bound'] = {'sphere': {'radius': 1.5}}
    m['rays'].update({'n_sample': 32, 'n_importance': 32, 'radius_bound': 1.5})
    m['chunk_pts'] = 4096
    m['geometry']['encoder'].update({'backend': 'torch', 'side': 3.0, 'n_levels': 8, 'hashmap_size': 12, 'base_res': 4, 'max_res': 64})
    m['radiance']['encoder']['view'].update({'backend': 'torch'})
    return yaml.dump(cfg, default_flow_style=False)


def main():
    text = edited_config()
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(text)
    torch.manual_seed(1818)
    model = build_model(load_configs(f.name, []), None)
    os.unlink(f.name)
    with torch.no_grad():   # the reference initialises the table in +-1e-4: give the encoder something to say
        emb = model.fg_model.geo_net.embed_fn.embeddings
        emb.copy_((torch.rand(emb.shape, generator=torch.Generator().manual_seed(1)) - 0.5) * 0.2)
    g = torch.Generator().manual_seed(1819)
    B, N = 2, 64
    o = torch.randn(B, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 3.0
    d = -o + (torch.rand(B, N, 3, generator=g) - 0.5) * 1.6
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(B, N, 1), 'img': torch.rand(B, N, 3, generator=g),
              'bkg_color': torch.rand(B, N, 3, generator=g)}
    out = {'config_yaml': np.array(text)}
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.detach().numpy()
    model.fg_model.set_ray_cfgs('perturb', False)
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    eik = ((res['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((res['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    loss.backward()
    out['train_loss'], out['train_eikonal'] = loss.detach().numpy(), eik.detach().numpy()
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
    path = os.path.join(OUT, 'g18_neus_ngp_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')
    print('hit rays', int((res['mask'] > 0).sum()), 'of', B * N, 'loss', float(loss), 'eik', float(eik), 'table grad max',
          float(model.fg_model.geo_net.embed_fn.embeddings.grad.abs().max()), 'params with grad',
          [k for k, p in model.named_parameters() if p.grad is not None])


if __name__ == '__main__':
    main()
