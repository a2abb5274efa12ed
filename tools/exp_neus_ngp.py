"""step time of the config-4 family at full size: NeuS on the hash grid inside the pruned volume (+ MultiVol background), torch Adam.
usage (GPU box): python tools/exp_neus_ngp.py [n_rays=4096] [bkg=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from arcnerf_amd.models import build_model
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

dev = torch.device('cuda:0')
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with_bkg = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
cfgs = load_configs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'neus_ngp_multivol.yaml'),
                    ['--model.rays.n_importance', '0'])
if not with_bkg:
    del cfgs.model.__dict__['background']
torch.manual_seed(0)
m = build_model(cfgs).to(dev)
vol = m.fg_model.obj_bound.volume
# an occupancy of ~5 % inside the side-1.5 volume
bf = torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev)
vol.update_bitfield(bf, ops='overwrite')
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=2.2)
inputs = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
          'bkg_color': torch.zeros(1, n_rays, 3, device=dev)}
tgt = torch.rand(1, n_rays, 3, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-15)


def step(it):
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
    loss = ((out['rgb'] - tgt) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return out


for it in range(3):
    out = step(it + 1)
torch.cuda.synchronize()
n_fg = int(out['normal_pts'].shape[0] * out['normal_pts'].shape[1])
t0 = time.perf_counter()
K = 10
for it in range(K):
    step(it + 10)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print('rays %d  dense fg sample slots %d  bkg %s  %.2f ms/step' % (n_rays, n_fg, with_bkg, dt * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for it in range(3):
        step(it + 30)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=18, max_name_column_width=70))
