"""host side of one module-path step of a bench config: python tools/exp_host_profile.py <config> [rays]   (GPU box)
torch-profiler table by self CPU time + host / wall ms per step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else 'neus_ngp_multivol'
spec = bench.MODULE_CONFIGS[name]
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', spec['yaml']), [])).to(dev)
fg = m.fg_model
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else spec['rays']
if name in ('neus_ngp_multivol', 'ngp_module'):
    fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
if name == 'neus_ngp_multivol':
    m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(m.bkg_model.n_grid, m.bkg_model.n_levels, 0.05, seed=5)).to(dev))
radius = 2.2 if name == 'neus_ngp_multivol' else (3.0 if name == 'neus' else (3.0 / 1.05 if name == 'ngp_module' else 4.0))
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=radius)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)}
if name == 'hdrnerf':
    inp['exp_time'] = torch.rand(1, n_rays, 1, device=dev) * 4.0 + 0.1
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def loss_of(out):
    if name == 'ngp_module':
        return torch.nn.functional.huber_loss(out['rgb_coarse'], inp['img'], delta=0.1)
    if name in ('nerf', 'hdrnerf'):
        return ((out['rgb_fine'] - inp['img']) ** 2).mean() + ((out['rgb_coarse'] - inp['img']) ** 2).mean()
    return ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()


def step(i):
    out = m({k: v for k, v in inp.items()}, inference_only=False, cur_epoch=20001 + i % 14)
    loss = loss_of(out)
    opt.zero_grad()
    loss.backward()
    opt.step()


for i in range(6):
    step(i)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('%s: host %.3f ms/step, wall %.3f ms/step' % (name, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=int(os.environ.get('ROWS', '60')), max_name_column_width=70))
