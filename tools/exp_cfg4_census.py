"""Where the launches of a config-4 step are: device-kernel count, device time and host time per phase.  GPU box."""
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
flat, b, n = m.prepare_flatten_inputs(inp)
phases = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        r = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_time_total > 0]
    phases[name] = (sum(e.count for e in ev), sum(e.device_time_total for e in ev) / 1e3, (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3)
    return r


def step(record):
    T = timed if record else (lambda n_, f: f())
    fg_out = T('fg forward', lambda: m.fg_model.forward(dict(flat), False, 't_last', 20000, 300000))
    bkg_out = T('bkg forward', lambda: m.bkg_model.forward(dict(flat), False, False, 20000, 300000))
    out = T('blend', lambda: m.reshape_output(m.detach_progress(m.blend_output(fg_out, bkg_out, False, False)), b, n))
    loss = T('loss', lambda: ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean())
    T('zero_grad', lambda: opt.zero_grad())
    T('backward', lambda: loss.backward())
    T('adam', lambda: opt.step())


for _ in range(3):
    step(False)
step(True)
print('%-14s %8s %10s %10s %10s' % ('phase', 'kernels', 'device ms', 'host ms', 'wall ms'))
for k, v in phases.items():
    print('%-14s %8d %10.3f %10.3f %10.3f' % ((k,) + v))
print('total kernels', sum(v[0] for v in phases.values()), 'device ms', sum(v[1] for v in phases.values()), 'host ms', sum(v[2] for v in phases.values()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(False)
torch.cuda.synchronize()
print('step: %.2f ms' % ((time.perf_counter() - t0) * 100))
