import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from arcnerf_amd import trainer as T
from arcnerf_amd.models import build_model
from arcnerf_amd.ops.multivol_func import multivol_rng
from arcnerf_amd.ops.volume_func import sampler_rng
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs
gpu = torch.device('cuda:0')
n_rays = 2048
loss_cfg = dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}})
g = torch.Generator().manual_seed(3)
pool = []
for i in range(8):
    o, d = synthetic_rays(n_rays, seed=20 + i, device=gpu, radius=2.2)
    pool.append({'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=gpu),
                 'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(gpu), 'img': torch.rand(1, n_rays, 3, generator=g).to(gpu)})
res = {}
for mode in (True, False, True):
    torch.manual_seed(0)
    m = build_model(load_configs('configs/neus_ngp_multivol.yaml', [])).to(gpu)
    m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(gpu), ops='overwrite')
    m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(gpu))
    with torch.no_grad():
        m.fg_model.geo_net.embed_fn.embeddings.mul_(200.0)
        m.bkg_model.geo_net.embed_fn.embeddings.mul_(2000.0)
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()
    sampler_rng(reset=True); multivol_rng(reset=True)
    m.train()
    st = T.FusedNeusNgpStep(m, T.build_loss(loss_cfg), opt, bkg_stream=mode)
    losses = []
    for i in range(400):
        _, l = st(pool[i % 8], 20000 + i, next_feed_in=[pool[(i + 1) % 8], pool[(i + 2) % 8]])
        if i % 50 == 49:
            losses.append(round(float(l['sum']), 5))
    torch.cuda.synchronize()
    p = opt.flat_params()
    print('bkg_stream', mode, losses, 'finite', bool(torch.isfinite(p).all()), 'norm', float(p.norm()))
