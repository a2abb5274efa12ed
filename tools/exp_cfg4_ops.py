"""config 4 (neus_ngp_multivol), one training step as bench.py runs it: the aten ops (CPU side) and device kernels per step, by count and
device time - where the torch fills / adds / copies between the arcn kernels come from.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(dev)
m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(dev))
n_rays = 4096
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=2.2)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)}
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def step():
    out = m(dict(inp), inference_only=False, cur_epoch=20000)
    loss = ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    if not getattr(opt, 'zero_grad_on_step', False):
        opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.device_time_total > 0 or e.key.startswith('aten::')]
# device kernels that are not ours, with the python frames that launched them
print('---- aten ops with device time, by (op, stack)')
agg = {}
for e in ka:
    if e.key.startswith('aten::') and e.self_device_time_total > 0:
        st = [s for s in e.stack if '/arcnerf_amd/' in s or 'exp_cfg4' in s or '/tests/' in s][:2]
        k = (e.key, ' <- '.join(s.split('/')[-1][:60] for s in st))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += e.count
        a[1] += e.self_device_time_total
for (op, st), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%-28s x%-3d %8.1f us  %s' % (op, c, t, st))
print('total aten self device time: %.1f us in %d launches' % (sum(v[1] for v in agg.values()), sum(v[0] for v in agg.values())))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print('step: %.3f ms' % ((time.perf_counter() - t0) * 50))
