"""One config-4 step (bench.py --config neus_ngp_multivol shapes) with every C-ABI call logged: entry point, rows.  GPU box."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd import _native as N
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(dev)
m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
n_rays = 4096
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=2.2)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)}
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def step():
    out = m({k: v for k, v in inp.items()}, inference_only=False, cur_epoch=20000)
    loss = ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
real = N.lib()
log = []


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith('arcn_') or name in ('arcn_last_error',):
            return fn

        def inner(*a):
            ints = [int(x) for x in a if isinstance(x, int) and 0 < x < (1 << 40)]
            log.append((name, ints[:6]))
            return fn(*a)
        return inner


N._lib = Proxy()
t0 = time.perf_counter()
step()
torch.cuda.synchronize()
print('step with logging: %.2f ms, %d C-ABI calls' % ((time.perf_counter() - t0) * 1e3, len(log)))
N._lib = real
cnt = collections.Counter(n for n, _ in log)
for n, c in cnt.most_common():
    print('%3d x %s   e.g. %s' % (c, n, [a for k, a in log if k == n][0]))
# torch-level profile: kernel launches per step
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages()
tot_cuda = sum(e.device_time_total for e in ev)
print('launches per step (device kernels): %d' % sum(e.count for e in ev if e.device_time_total > 0 and e.cpu_time_total == 0))
for e in sorted(ev, key=lambda e: -e.device_time_total)[:30]:
    if e.device_time_total > 0:
        print('%8.1f us %4d x  %s' % (e.device_time_total, e.count, e.key[:100]))
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print('step: %.2f ms' % ((time.perf_counter() - t0) * 100))
