"""config 4 through trainer.FusedNeusNgpStep: where the HOST time of a step goes (cProfile over 60 steps, the GPU never the limiter: the
profile is taken with the device running ahead... it cannot - the step waits for the prefetched totals - so the wall time per step is printed
next to the profile).  GPU box."""
import cProfile, io, os, pstats, sys, time
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
n_rays = int(os.environ.get('N_RAYS', '4096'))
pool = []
for k in range(4):
    o, d = synthetic_rays(n_rays, seed=k, device=dev, radius=2.2)
    pool.append({'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
                 'bkg_color': torch.rand(1, n_rays, 3, device=dev), 'img': torch.rand(1, n_rays, 3, device=dev)})
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15, zero_grad_on_step=True).flatten()
loss = T.build_loss(dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}}))
m.train()
stepper = T.FusedNeusNgpStep(m, loss, opt)


AHEAD = int(os.environ.get('AHEAD', '2'))
STALL_MS = float(os.environ.get('STALL_MS', '0'))      # a host that disappears for this long before every 4th step (a busy shared machine)


def run(lo, hi):
    for i in range(lo, hi):
        if STALL_MS > 0 and i % 4 == 0:
            time.sleep(STALL_MS * 1e-3)
        stepper(pool[i % 4], 20000 + i, next_feed_in=[pool[(i + k) % 4] for k in range(1, AHEAD + 1)])


run(0, 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(16, 76)
torch.cuda.synchronize()
print('wall per step: %.3f ms (%d rays, %d batches ahead, host stall %.1f ms every 4th step)' % ((time.perf_counter() - t0) / 60 * 1e3, n_rays, AHEAD, STALL_MS))
if os.environ.get('NO_PROFILE'):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
run(76, 136)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print('\n'.join(l[:170] for l in s.getvalue().split('\n')[:50]))
