"""Where a recorded module step spends its time: the replay alone (host issue / device), and the eager pieces around it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
from arcnerf_amd.trainer import GraphedTrainStep
from arcnerf_amd.utils.cfgs_utils import load_configs
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), [])).to(dev)
fg = m.fg_model
fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(dev), ops='overwrite')
n_rays = 8320
o, d = synthetic_rays(n_rays, seed=0, device=dev, radius=3.0 / 1.05)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
       'bkg_color': torch.rand(1, n_rays, 3).to(dev), 'img': torch.rand(1, n_rays, 3).to(dev)}
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()
gs = GraphedTrainStep(m, lambda i, out: {'sum': torch.nn.functional.huber_loss(out['rgb_coarse'], i['img'], delta=0.1)}, opt)
for k in range(6):
    gs(dict(inp), k)
torch.cuda.synchronize()
g = list(gs.graphs.values())[0][0]
for name, fn in (('replay only', g.replay), ('full call', lambda: gs(dict(inp), 7))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print('%s: %.3f ms per step (host issue %.3f ms)' % (name, t / 30 * 1e3, th / 30 * 1e3))
if len(sys.argv) > 1:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
ts = []
for k in range(150):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gs(dict(inp), 100 + k)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print('per-call ms, synchronised:', [round(v, 2) for v in ts[::10]])
t0 = time.perf_counter()
for k in range(100):
    gs(dict(inp), 300 + k)
torch.cuda.synchronize()
print('100 calls back to back: %.3f ms per step' % ((time.perf_counter() - t0) * 10))
pool = []
gq = torch.Generator(device='cpu').manual_seed(77)
for i in range(4):
    o2, d2 = synthetic_rays(n_rays, seed=i, device=dev, radius=3.0 / 1.05)
    pool.append({'rays_o': o2.view(1, -1, 3), 'rays_d': d2.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
                 'bkg_color': torch.rand(1, n_rays, 3, generator=gq).to(dev), 'img': torch.rand(1, n_rays, 3, generator=gq).to(dev)})
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(100):
    gs({kk: v for kk, v in pool[k % 4].items()}, 500 + k)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('100 calls over a pool of 4 batches: %.3f ms per step (host %.3f), graphs %d' % ((time.perf_counter() - t0) * 10, th * 10, len(gs.graphs)))
for rep in range(3):
    gs.host_s.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        gs({kk: v for kk, v in pool[k % 4].items()}, 600 + k)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('pool, 100 calls, round %d: %.3f ms per step (host %.3f) %s' % (rep, (time.perf_counter() - t0) * 10, th * 10, {k: round(v * 10, 3) for k, v in gs.host_s.items()}))
