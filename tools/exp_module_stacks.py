"""Which source lines of arcnerf_amd launch the non-arcn kernels of a module-path step: torch.profiler with_stack, device time of
aten::{cat,copy_,fill_,add_,add,mul,div,zero_,...} grouped by the innermost arcnerf_amd frame.  usage: python tools/exp_module_stacks.py nerf|neus|hdrnerf|neus_ngp_multivol"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

name = sys.argv[1] if len(sys.argv) > 1 else 'nerf'
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', name + '.yaml'), [])).to(dev)
n_rays = 2048 if name == 'neus' else 4096
radius = 2.2 if name == 'neus_ngp_multivol' else (3.0 if name == 'neus' else 4.0)
if name == 'neus_ngp_multivol':
    m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
    m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(dev))
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=radius)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)}
if name == 'hdrnerf':
    inp['exp_time'] = torch.rand(1, n_rays, 1, device=dev) * 4.0 + 0.1
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def step():
    out = m(dict(inp), inference_only=False, cur_epoch=20000)
    if 'normal_pts' in out:
        loss = ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    else:
        loss = ((out['rgb_fine'] - inp['img']) ** 2).mean() + ((out['rgb_coarse'] - inp['img']) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith('aten::') or e.self_device_time_total <= 0:
        continue
    frame = 'autograd engine / no python frame'
    for fr in (e.stack or []):
        if 'arcnerf_amd' in fr or 'tools/' in fr or 'bench.py' in fr:
            frame = fr.split('arcnerf_amd/')[-1] if 'arcnerf_amd/' in fr else fr
            break
    k = (e.name, frame[:110])
    by[k][0] += e.self_device_time_total
    by[k][1] += 1
tot = sum(v[0] for v in by.values())
print('%s: aten ops with their own kernels: %.2f ms' % (name, tot / 1e3))
for (op, fr), (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:45]:
    print('%8.1f us %4d x  %-22s %s' % (t, c, op, fr))
