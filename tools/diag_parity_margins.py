"""How far the full-width model fixtures (G22 / G23 / G24) and G18 are from their bars: max |err| / (atol + rtol |ref|) per output key at
rtol = atol = 1e-4, and the gradient errors relative to each tensor's max.  GPU box."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import seeded_weights as SW
from arcnerf_amd.models import build_model
from arcnerf_amd.utils.cfgs_utils import load_configs

gpu = torch.device('cuda:0')


def ratio(a, b, tol=1e-4):
    return float((np.abs(a - b) / (tol + tol * np.abs(b))).max())


def run(cfg, fixture, neus=False):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', fixture + '.npz'))
    m = build_model(load_configs(os.path.join(ROOT, 'configs', cfg + '.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in SW.state_dict_from_fixture(g).items()})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    print(fixture, 'infer', {k: round(ratio(out[k].detach().cpu().numpy(), g['infer_' + k]), 3) for k in out if 'infer_' + k in g.files})
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    print(fixture, 'train', {k: round(ratio(out[k].detach().cpu().numpy(), g['train_' + k]), 3) for k in out
                             if torch.is_tensor(out[k]) and 'train_' + k in g.files})


if __name__ == '__main__':
    run('nerf', 'g22_nerf_fullwidth')
    run('neus', 'g23_neus_fullwidth', True)
    run('hdrnerf', 'g24_hdrnerf_fullwidth')
