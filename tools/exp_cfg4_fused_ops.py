"""config 4 (neus_ngp_multivol) through trainer.FusedNeusNgpStep, one step under the torch profiler: every aten op with device time and the
python line of the package that issued it, the device kernels by count and time, and the arcn:: share.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd import trainer as T
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs

dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(dev)
m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(dev))
n_rays = 4096
pool = []
for k in range(4):
    o, d = synthetic_rays(n_rays, seed=k, device=dev, radius=2.2)
    pool.append({'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
                 'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)})
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15, zero_grad_on_step=True).flatten()
loss = T.build_loss(dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}}))
m.train()
stepper = T.FusedNeusNgpStep(m, loss, opt)


def step(i):
    return stepper(pool[i % 4], 20000 + i, next_feed_in=pool[(i + 1) % 4])


for i in range(8):
    step(i)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(8, 12):
        step(i)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=8)
agg = {}
for e in ka:
    if e.key.startswith('aten::') and e.self_device_time_total > 0:
        st = [s for s in e.stack if '/arcnerf_amd/' in s][:2]
        k = (e.key, ' <- '.join(s.split('/arcnerf_amd/')[-1][:70] for s in st))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += e.count
        a[1] += e.self_device_time_total
print('---- aten ops with device time (4 steps), by (op, first package frames)')
for (op, st), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-26s x%-3d %7.1f us  %s' % (op, c, t, st))
print('total aten self device time per step: %.1f us in %.1f launches' % (sum(v[1] for v in agg.values()) / 4, sum(v[0] for v in agg.values()) / 4))
kern = {}
for e in prof.key_averages():
    if e.device_type == torch.autograd.DeviceType.CUDA and e.self_device_time_total > 0 and not e.key.startswith('aten::'):
        kern[e.key] = (e.count, e.self_device_time_total)
tot = sum(v[1] for v in kern.values())
ours = sum(v[1] for k, v in kern.items() if 'arcn::' in k)
print('---- device kernels: %.1f us per step in %.1f launches, arcn:: %.1f %%' % (tot / 4, sum(v[0] for v in kern.values()) / 4, 100 * ours / max(tot, 1e-9)))
for k, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
    if 'arcn::' not in k:
        print('%-100s x%-3d %7.1f us' % (k[:100], c, t))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12, 52):
    step(i)
torch.cuda.synchronize()
print('step: %.3f ms' % ((time.perf_counter() - t0) * 25))

# ---- who issues them: every aten op of one step that launches something, with the package frames it came from
import traceback
from collections import Counter
from torch.utils._python_dispatch import TorchDispatchMode


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ('view', 'reshape', 'slice', 'select', 'as_strided', 'detach', 'empty', 'expand', 'unsqueeze', 'squeeze', 'transpose',
                                        'permute', ' t.default', 'alias', 'record_stream', 'is_pinned', 'item', '_local_scalar', 'unbind', 'split', 'narrow', 'size', 'stride')):
            fr = [f for f in traceback.extract_stack() if '/arcnerf_amd/' in f.filename][-3:]
            self.c[(name, ' <- '.join('{}:{}'.format(f.filename.split('/arcnerf_amd/')[-1], f.lineno) for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


with Log() as lg:
    step(60)
torch.cuda.synchronize()
print('---- aten calls of one step (views and allocations left out)')
for (name, where), c in sorted(lg.c.items(), key=lambda kv: kv[0][1]):
    print('%-34s x%-2d %s' % (name.replace('aten.', ''), c, where))
