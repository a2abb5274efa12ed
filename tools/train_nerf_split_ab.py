"""Does the split-bf16 form of the dense layers train like the exact-f32 form?  configs/nerf.yaml (vanilla NeRF, 8 x 256 nets, coarse +
fine) on the analytic scene of tools/psnr_curve.py, 1024 rays per step, same seed, same batches; held-out PSNR at a few iterations.
usage (GPU box): ARCN_GEMM_SPLIT=1 python tools/train_nerf_split_ab.py [max_iter=1500];  ARCN_GEMM_SPLIT=0 python ... (A/B)"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = sys.argv[:1] + ['0'], sys.argv
import torch

MAX_IT = int(argv[1]) if len(argv) > 1 else 1500
import importlib.util
spec = importlib.util.spec_from_file_location('psnr_scene', os.path.join(ROOT, 'tools', 'psnr_curve.py'))
scene = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scene)
train, test, dev = scene.train, scene.test, scene.dev

from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.ops import functional as F
from arcnerf_amd.utils.cfgs_utils import load_configs

torch.manual_seed(0)
m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf.yaml'), ['--model.rays.white_bkg', 'True', '--model.rays.near', '1.2', '--model.rays.far', '4.6'])).to(dev)
opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-8)


@torch.no_grad()
def psnr():
    mse, n = 0.0, 0
    for o, d, tgt in test[:1]:
        for c in range(0, 8192, 4096):
            out = m({'rays_o': o[None, c:c + 4096], 'rays_d': d[None, c:c + 4096], 'rays_r': torch.zeros(1, 4096, 1, device=dev),
                     'bkg_color': torch.ones(1, 4096, 3, device=dev)}, inference_only=True)
            mse += float(((out['rgb'][0] - tgt[c:c + 4096]) ** 2).sum())
            n += 4096 * 3
    return -10.0 * math.log10(mse / n)


n_rays, points = 1024, []
t0 = time.perf_counter()
for it in range(1, MAX_IT + 1):
    o, d, tgt, _ = train[it % len(train)]
    inputs = {'rays_o': o[None, :n_rays], 'rays_d': d[None, :n_rays], 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
              'bkg_color': torch.ones(1, n_rays, 3, device=dev)}
    out = m(inputs, inference_only=False, cur_epoch=it)
    loss = ((out['rgb_fine'][0] - tgt[:n_rays]) ** 2).mean() + ((out['rgb_coarse'][0] - tgt[:n_rays]) ** 2).mean()
    opt.zero_grad(set_to_none=False)
    loss.backward()
    opt.step()
    if it in (100, 500, 1000, MAX_IT):
        torch.cuda.synchronize()
        points.append({'iter': it, 'psnr': round(psnr(), 3), 'loss': float(loss), 'train_seconds': round(time.perf_counter() - t0, 1)})
        print(json.dumps(points[-1]), file=sys.stderr, flush=True)
print(json.dumps({'path': 'build_model(nerf.yaml) + FusedAdam, 1024 rays/step', 'split_products': bool(F._GEMM_SPLIT), 'points': points}))
