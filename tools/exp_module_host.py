"""host side of one module-path NGP step (bench.py --config ngp_module shapes): cProfile by own time, and the torch ops issued per step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.utils.cfgs_utils import load_configs
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), [])).to(dev)
m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)).to(dev), ops='overwrite')
R = 8320
o, d = synthetic_rays(R, seed=0, device=dev)
inp = {'rays_o': o.view(1, R, 3), 'rays_d': d.view(1, R, 3), 'rays_r': torch.zeros(1, R, 1, device=dev),
       'img': torch.rand(1, R, 3, device=dev), 'bkg_color': torch.rand(1, R, 3, device=dev)}
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()


def step(i):
    out = m({k: v for k, v in inp.items()}, inference_only=False, cur_epoch=20001 + i % 14)
    loss = torch.nn.functional.huber_loss(out['rgb_coarse'], inp['img'], delta=0.1)
    opt.zero_grad()
    loss.backward()
    opt.step()


for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host %.3f ms/step, wall %.3f ms/step' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=45, max_name_column_width=60))

# which source lines issue the torch ops
import collections
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof2:
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
by = collections.Counter()
for e in prof2.events():
    if not e.name.startswith('aten::') or e.name in ('aten::view', 'aten::slice', 'aten::as_strided', 'aten::empty', 'aten::empty_strided', 'aten::select', 'aten::narrow', 'aten::empty_like', 'aten::resize_', 'aten::reshape', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze', 'aten::detach', 'aten::alias', 'aten::_unsafe_view', 'aten::permute', 'aten::transpose', 'aten::t'):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue
    fr = [s for s in (e.stack or []) if '/root/' in s or 'repo' in s]
    by[(e.name, fr[0] if fr else '?')] += 1
for (n, f), c in sorted(by.items(), key=lambda kv: -kv[1]):
    print('%3d x %-22s %s' % (c // 3, n, f[-110:]))

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(40)
