"""which torch ops (with shapes / call sites) make the elementwise and copy kernels of a module-path step: tools/exp_copy_sites.py nerf"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_rays
from arcnerf_amd.utils.cfgs_utils import load_configs
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'nerf'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
torch.manual_seed(0)
m = build_model(load_configs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', name + '.yaml'), [])).to(dev)
o, d = synthetic_rays(R, seed=0, device=dev, radius=3.0 if name == 'neus' else 4.0)
inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, R, 1, device=dev), 'bkg_color': torch.rand(1, R, 3, device=dev)}
img = torch.rand(1, R, 3, device=dev)
if name == 'hdrnerf':
    inp['exp_time'] = torch.rand(1, R, 1, device=dev) * 4.0 + 0.1
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15)
if name == 'nerf_ngp':
    from arcnerf_amd.pipeline import synthetic_bitfield
    fgm = m.fg_model
    if hasattr(fgm, 'obj_bound') and hasattr(fgm.obj_bound, 'volume'):
        pass
def step(i):
    out = m(dict(inp), inference_only=False, cur_epoch=20000 + i)
    if name.startswith('neus'):
        loss = ((out['rgb'] - img) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    elif 'rgb_fine' in out:
        loss = ((out['rgb_fine'] - img) ** 2).mean() + ((out['rgb_coarse'] - img) ** 2).mean()
    else:
        loss = ((out[[k for k in out if k.startswith('rgb')][0]] - img) ** 2).mean()
    opt.zero_grad(set_to_none=False); loss.backward(); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(10)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=42, max_shapes_column_width=70))
