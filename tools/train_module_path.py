"""The drop-in path end to end: configs/nerf_ngp.yaml -> build_model -> FullModel.forward / optimize -> FusedAdam, trained on the
analytic scene of tools/psnr_curve.py (same rays, same ground truth) - what a reference user gets after swapping the package, with
the reference trainer's cadence: optimize() every step (the bound refreshes every 16), dynamic batch size from
get_dynamicbs_factor() every 16 steps, MultiStepLR.  usage (GPU box): python tools/train_module_path.py [max_iter=4000]"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = sys.argv[:1] + ['0'], sys.argv   # psnr_curve builds its data at import; skip its own training loop
import torch

MAX_IT = int(argv[1]) if len(argv) > 1 else 4000
import importlib.util
spec = importlib.util.spec_from_file_location('psnr_scene', os.path.join(ROOT, 'tools', 'psnr_curve.py'))
scene = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scene)      # MAX_IT = 0 there: data + pipeline objects only
train, test, dev = scene.train, scene.test, scene.dev

from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.utils.cfgs_utils import load_configs

m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), ['--model.rays.white_bkg', 'True'])).to(dev)
fg = m.fg_model
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-1, eps=1e-15, weight_decay=1e-6, ema_decay=0.95,
                zero_grad_on_step=True, ema_in_param=True)   # optim block of nerf_lego_nerf_ngp.yaml: Adam 1e-1 WITH the EMA write-back (ema.decay 0.95)


@torch.no_grad()
def psnr():
    mse, n = 0.0, 0
    for o, d, tgt in test:
        out = m({'rays_o': o[None], 'rays_d': d[None], 'rays_r': torch.zeros(1, o.shape[0], 1, device=dev),
                 'bkg_color': torch.ones(1, o.shape[0], 3, device=dev)}, inference_only=True)
        mse += float(((out['rgb'][0] - tgt) ** 2).sum())
        n += tgt.numel()
    return -10.0 * math.log10(mse / n)


n_rays, points, t_train = 4096, [], 0.0
t_last = time.perf_counter()
for it in range(1, MAX_IT + 1):
    o, d, tgt, _ = train[it % len(train)]
    inputs = {'rays_o': o[None, :n_rays], 'rays_d': d[None, :n_rays], 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
              'bkg_color': torch.ones(1, n_rays, 3, device=dev)}
    out = m(inputs, inference_only=False, cur_epoch=it)
    loss = torch.nn.functional.huber_loss(out['rgb_coarse'][0], tgt[:n_rays], delta=0.1) * (3000.0 / 0.1)   # the reference's Huber is torch's / delta (loss/img_loss.py:80-100)
    loss.backward()
    opt.step()
    m.optimize(cur_epoch=it)
    if it % 16 == 0:   # the reference's dynamic batch size (trainer/pipeline.py:222-241)
        n_rays = int(min(32768, max(128, (int(n_rays * fg.get_dynamicbs_factor()) + 127) // 128 * 128)))
    if it in (100, 500, 2000, MAX_IT):
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t_last
        occ = float(fg.obj_bound.volume.get_voxel_bitfield().float().mean())
        points.append({'iter': it, 'psnr': psnr(), 'loss': float(loss), 'train_seconds': t_train, 'occupied': occ, 'rays_per_step': n_rays})
        print(json.dumps(points[-1]), file=sys.stderr, flush=True)
        t_last = time.perf_counter()
print(json.dumps({'path': 'build_model(nerf_ngp.yaml) + FusedAdam', 'points': points}))
