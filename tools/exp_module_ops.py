"""One training step of a module-path config under torch.profiler: aten ops / autograd nodes by device time, with call counts (where do the
non-arcn kernels of a step come from?).  usage (GPU box): python tools/exp_module_ops.py nerf|neus|hdrnerf [chunk_pts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

name = sys.argv[1] if len(sys.argv) > 1 else 'nerf'
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', name + '.yaml'), [])).to(dev)
if len(sys.argv) > 2:
    m.set_chunk_pts(int(sys.argv[2]))
n_rays = 2048 if name == 'neus' else 4096
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=3.0 if name == 'neus' else 4.0)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)}
if name == 'hdrnerf':
    inp['exp_time'] = torch.rand(1, n_rays, 1, device=dev) * 4.0 + 0.1
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def step():
    out = m(dict(inp), inference_only=False, cur_epoch=20000)
    if name == 'neus':
        loss = ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    else:
        loss = ((out['rgb_fine'] - inp['img']) ** 2).mean() + ((out['rgb_coarse'] - inp['img']) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [e for e in ev if e.device_time_total > 0 and e.cpu_time_total == 0]
tot = sum(e.device_time_total for e in kern)
arcn = sum(e.device_time_total for e in kern if 'arcn' in e.key)
print('%s: %d launches, kernel time %.2f ms, arcn share %.1f %%' % (name, sum(e.count for e in kern), tot / 1e3, 100 * arcn / tot))
print('--- non-arcn kernels')
for e in sorted([e for e in kern if 'arcn' not in e.key], key=lambda e: -e.device_time_total)[:14]:
    print('%8.1f us %5d x  %s' % (e.device_time_total, e.count, e.key[:110]))
print('--- aten ops / autograd nodes (self device time of their kernels)')
ops = [e for e in ev if e.cpu_time_total > 0 and e.self_device_time_total > 0]
for e in sorted(ops, key=lambda e: -e.self_device_time_total)[:30]:
    print('%8.1f us %5d x  %s' % (e.self_device_time_total, e.count, e.key[:90]))
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print('step: %.2f ms' % ((time.perf_counter() - t0) * 200))
