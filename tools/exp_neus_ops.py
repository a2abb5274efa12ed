"""config 3 (neus.yaml, the wide sdf net with normals + Eikonal) as bench.py runs it: the aten ops of one step with the package line that
issued them (forward and the hand-written backward of ops/sdf_chain.py; C++ autograd nodes show no frame) and their device time by kernel.  GPU box."""
import os, sys, time, traceback
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus.yaml'), [])).to(dev)
n_rays = 2048
pool = []
g = torch.Generator().manual_seed(77)
for k in range(4):
    o, d = synthetic_rays(n_rays, seed=k, device=dev, radius=3.0)
    pool.append({'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
                 'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(dev), 'img': torch.rand(1, n_rays, 3, generator=g).to(dev)})
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15, zero_grad_on_step=True).flatten()
m.train()


def step(i):
    inp = pool[i % 4]
    out = m(dict(inp), inference_only=False, cur_epoch=20000 + i)
    loss = ((out['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss.backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for i in range(4, 6):
        step(i)
    torch.cuda.synchronize()
kern = {}
for e in prof.key_averages():
    if e.device_type == torch.autograd.DeviceType.CUDA and e.self_device_time_total > 0 and not e.key.startswith('aten::'):
        kern[e.key] = (e.count, e.self_device_time_total)
tot = sum(v[1] for v in kern.values())
ours = sum(v[1] for k, v in kern.items() if 'arcn::' in k)
print('---- device kernels: %.1f us per step in %.1f launches, arcn:: %.1f %%' % (tot / 2, sum(v[0] for v in kern.values()) / 2, 100 * ours / max(tot, 1e-9)))
for k, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
    if 'arcn::' not in k:
        print('%-110s x%-4d %8.1f us/step' % (k[:110], c // 2, t / 2))
agg = {}
for e in prof.key_averages():
    if e.key.startswith('aten::') and e.self_device_time_total > 0:
        agg[e.key] = (e.count, e.self_device_time_total)
print('---- aten ops with device time per step')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-28s x%-4d %8.1f us' % (k, c // 2, t / 2))

SKIP = ('view', 'reshape', 'slice', 'select', 'as_strided', 'detach', 'empty', 'expand', 'unsqueeze', 'squeeze', 'transpose', 'permute', ' t.default', 'alias',
        'record_stream', 'is_pinned', 'item', '_local_scalar', 'unbind', 'split', 'narrow', 'size', 'stride', '_unsafe_view')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            fr = [f for f in traceback.extract_stack() if '/arcnerf_amd/' in f.filename][-3:]
            numel = 0
            for a in args:
                if torch.is_tensor(a):
                    numel = max(numel, a.numel())
            self.c[(name, ' <- '.join('{}:{}'.format(f.filename.split('/arcnerf_amd/')[-1], f.lineno) for f in reversed(fr)))] += 1
            self.c[('~numel', name + ' ' + ' <- '.join('{}:{}'.format(f.filename.split('/arcnerf_amd/')[-1], f.lineno) for f in reversed(fr[-1:])))] = numel
        return func(*args, **(kwargs or {}))


with Log() as lg:
    step(60)
torch.cuda.synchronize()
print('---- aten calls of one step (views and allocations left out), with the largest operand')
sizes = {k[1]: v for k, v in lg.c.items() if k[0] == '~numel'}
for (name, where), c in sorted(((k, v) for k, v in lg.c.items() if k[0] != '~numel'), key=lambda kv: kv[0][1]):
    first = where.split(' <- ')[0]
    print('%-30s x%-3d n=%-10d %s' % (name.replace('aten.', ''), c, sizes.get(name + ' ' + first, 0), where))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12, 32):
    step(i)
torch.cuda.synchronize()
print('step: %.3f ms' % ((time.perf_counter() - t0) * 50))
