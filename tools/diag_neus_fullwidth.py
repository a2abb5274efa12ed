import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import seeded_weights as SW
from arcnerf_amd.models import build_model
from arcnerf_amd.utils.cfgs_utils import load_configs
g = np.load('/root/repo/tests/golden/g23_neus_fullwidth.npz')
gpu = torch.device('cuda:0')
m = build_model(load_configs('/root/repo/configs/neus.yaml', [str(v) for v in g['overrides']])).to(gpu)
m.load_state_dict({k: torch.from_numpy(v) for k, v in SW.state_dict_from_fixture(g).items()})
inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
m.fg_model.set_ray_cfgs('noise_std', 0.0)
from rand_feed import RandFeed
seen_z, real_mid = [], m.fg_model.handle_mid_pts
m.fg_model.handle_mid_pts = lambda z, mk: (seen_z.append(z.detach().clone()), real_mid(z, mk))[1]
with RandFeed([g[k] for k in sorted(k for k in g.files if k.startswith('draw_'))], gpu):
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
if 'train_zvals_upsampled' in g.files:
    dz = np.abs(seen_z[0].cpu().numpy() - g['train_zvals_upsampled']); print('zvals max diff', dz.max(), 'rays > 1e-5', np.unique(np.nonzero(dz > 1e-5)[0]), 'count > 1e-6', (dz > 1e-6).sum())
npts = out['normal_pts'].detach().cpu().numpy(); ref = g['train_normal_pts']
bad = np.abs(npts - ref) > 2e-4 + 2e-4*np.abs(ref)
print('normal_pts bad frac', bad.mean(), 'rows with bad', np.unique(np.nonzero(bad)[0]).size, 'max diff', np.abs(npts-ref).max())
eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
loss = ((out['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
loss.backward()
for n, p in m.named_parameters():
    if ('gsum.' + n + '.max') in g.files:
        gg = p.grad.cpu().numpy().reshape(p.shape[0], -1)
        mx = float(g['gsum.' + n + '.max'])
        e1 = np.abs(gg[:4] - g['gsum.' + n + '.head']).max() / mx
        e2 = np.abs(gg[5::16] - g['gsum.' + n + '.mod16']).max() / mx
        print('%-50s head %.2e mod16 %.2e' % (n, e1, e2))
    elif ('grad.' + n) in g.files:
        r = g['grad.' + n]; print('%-50s full %.2e' % (n, np.abs(p.grad.cpu().numpy() - r).max() / (np.abs(r).max() + 1e-12)))
