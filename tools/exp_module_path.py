"""experiment: throughput of the drop-in nn.Module path (reference YAML -> build_model -> autograd -> torch.optim.Adam)
on the bench workload, next to the packed pipeline"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from arcnerf_amd.models import build_model
from arcnerf_amd.utils.cfgs_utils import load_configs
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
dev = torch.device('cuda:0')
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), [])).to(dev)
fg = m.fg_model
fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, 0)).to(dev), ops='overwrite')
R = 8320
o, d = synthetic_rays(R, seed=0, device=dev)
inputs = {'rays_o': o.view(1, R, 3), 'rays_d': d.view(1, R, 3), 'rays_r': torch.zeros(1, R, 1, device=dev),
          'img': torch.rand(1, R, 3, device=dev), 'bkg_color': torch.ones(1, R, 3, device=dev)}
tgt = torch.rand(1, R, 3, device=dev)
if os.environ.get('FUSED_ADAM', '0') == '1':
    from arcnerf_amd.optim import FusedAdam
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, eps=1e-15, zero_grad_on_step=True)
else:
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-15)
def step(it):
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
    loss = torch.nn.functional.huber_loss(out['rgb_coarse'], tgt, delta=0.1)
    if os.environ.get('FUSED_ADAM', '0') != '1':
        opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    return loss
for it in range(5):
    step(it)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for it in range(N):
    step(1000 + it)   # away from the refresh cadence? (cur_epoch % 16 decides) - includes refreshes like training does
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print('module path: %.3f ms/step' % (dt * 1e3))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for it in range(10):
        step(2000 + it)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
