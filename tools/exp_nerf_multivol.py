"""step time of configs/nerf_multivol.yaml at full size: packed NGP foreground + MultiVol background (hash grid + fused MLPs over
the 5-level cascade), rgb blending, FusedAdam.  usage (GPU box): python tools/exp_nerf_multivol.py [n_rays=4096]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

dev = torch.device('cuda:0')
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_multivol.yaml'), [])).to(dev)
m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
bkg = m.bkg_model
with torch.no_grad():   # a sparse background occupancy (about 3 % of every level)
    bits = (torch.rand(bkg.total_n_elements, device=dev) < 0.03).view(-1, 8)
    w = (2 ** torch.arange(8, device=dev)).to(torch.int32)
    bkg.density_bitfield.copy_((bits.to(torch.int32) * w).sum(-1).to(torch.uint8))
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=0.45)   # cameras inside the inner volume looking outwards too
inputs = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
          'bkg_color': torch.zeros(1, n_rays, 3, device=dev)}
tgt = torch.rand(1, n_rays, 3, device=dev)
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15, zero_grad_on_step=True)


def step(it):
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
    loss = ((out['rgb_coarse'] - tgt) ** 2).mean()
    loss.backward()
    opt.step()


for it in range(3):
    step(it + 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for it in range(K):
    step(it + 17)
torch.cuda.synchronize()
print('rays %d  %.2f ms/step' % (n_rays, (time.perf_counter() - t0) / K * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for it in range(3):
        step(it + 33)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=64))
