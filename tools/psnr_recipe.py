"""PSNR@iter with the reference's RECIPE through the drop-in API (the metric's second half: BASELINE.json "... + PSNR@iter").

The loop is the one golden G27 pins to the reference (tests/test_gpu_psnr.py, 100 x 100 images): `build_model(configs/nerf_ngp.yaml)` with
the model block of configs/expr/NeRF/lego/nerf_lego_nerf_ngp.yaml + `trainer.train_epoch` + `trainer.FusedNgpStep` (two batches in flight)
+ `trainer.TrainBatches` on a `trainer.Pipeline` with that yaml's dataset scheduler - centre precrop 0.5 for the first 500 iterations,
random background colours blended into the targets, cross-view shuffle, dynamic batch size every 16 iterations from 4096 rays - FusedAdam
1e-1 / eps 1e-15 / weight decay 1e-6, EMA 0.95 written back, MultiStepLR 0.33 @ 20k/30k/40k/50k.  Here at the size the box allows with no
dataset on it: the analytic scene of tools/psnr_curve.py (six soft textured blobs, mildly view dependent) rendered ONCE on the GPU to 100
training views of 320 x 320 RGBA BYTES (what a Blender dataset holds: straight colours + alpha; 41 MB in HBM - the reference's per-pixel
tensors of the same views would be 450 MB) and 4 held-out views; PSNR = -10 log10(mse) over all held-out pixels on white
(img_metric.py:50-56), through the module's own inference (`model(..., inference_only=True)` in eval mode, EMA parameters).
320 x 320 because the recipe needs it: 100 views x 160 x 160 cropped rays / 4096 = 625 batches > precrop.max_epoch 500 - with fewer
cropped rays the reshuffle of a finished pass comes first and the reference's state machine never ends the crop.

usage (GPU box): python tools/psnr_recipe.py [max_iter=2000] [seeds=0]      -> JSON
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from arcnerf_amd import trainer as T
from arcnerf_amd.models import build_model
from arcnerf_amd.ops import functional as F
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.render.ray_helper import get_rays
from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs

HW, N_TRAIN, N_TEST = 320, 100, 4
ANGLE, RADIUS = 0.6911, 3.0 / 1.05
EXPR_MODEL = ['--model.rays.noise_std', '0.0', '--model.rays.white_bkg', 'True', '--model.obj_bound.bkg_color', '[1.0,1.0,1.0]']
SCHEDULER = {'precrop': {'ratio': 0.5, 'max_epoch': 500}, 'bkg_color': {'color': 'random'}, 'dynamic_batch_size': {'update_epoch': 16}}
_SCENE = {}


def _cameras(first, n, dev):
    Ks, Ms = [], []
    for v in range(first, first + n):
        gg = torch.Generator(device='cpu').manual_seed(v)
        c = torch.randn(3, generator=gg)
        c = c / c.norm() * RADIUS
        fwd = -c / c.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, -up, fwd, c
        focal = 0.5 * HW / math.tan(0.5 * ANGLE)
        Ks.append(torch.tensor([[focal, 0.0, HW / 2], [0.0, focal, HW / 2], [0.0, 0.0, 1.0]]))
        Ms.append(c2w)
    return torch.stack(Ks).to(dev), torch.stack(Ms).to(dev)


def build_image_scene(dev=None):
    """-> dict(rgba_train (N, HW*HW, 4) uint8, K / c2w of the training and held-out views, test targets on white (4, HW*HW, 3), seconds)"""
    dev = dev or torch.device('cuda:0')
    if dev in _SCENE:
        return _SCENE[dev]
    rng = np.random.default_rng(3)
    centers = torch.tensor((rng.random((6, 3)) - 0.5) * 1.0, dtype=torch.float32, device=dev)
    radii = torch.tensor(rng.random(6) * 0.18 + 0.12, dtype=torch.float32, device=dev)
    phase = torch.tensor(rng.random((6, 3)) * 6.28, dtype=torch.float32, device=dev)

    def field(x, d):
        r2 = ((x[:, None, :] - centers[None]) ** 2).sum(-1) / (radii[None] ** 2)
        w = torch.sigmoid((1.0 - r2) * 12.0)
        sigma = 60.0 * w.max(dim=1)[0]
        base = 0.5 + 0.5 * torch.sin(phase[None] + 4.0 * x[:, None, :])
        col = (w[..., None] * base).sum(1) / (w.sum(1, keepdim=True) + 1e-6)
        col = (col * (0.75 + 0.25 * (d * x).sum(-1, keepdim=True).tanh())).clamp(0, 1)
        return sigma, col

    @torch.no_grad()
    def render(K, c2w, n=768, chunk=8192):
        o, d, _, _ = get_rays(HW, HW, K, c2w, wh_order=False, center_pixel=True)
        out = torch.zeros(o.shape[0], 4, device=dev)
        # only rays that come within 1.6 radii of a blob centre can see density (w < 1e-8 beyond): the others stay transparent
        tc = ((centers[None] - o[:, None]) * d[:, None]).sum(-1)
        dist2 = ((o[:, None] + tc[..., None] * d[:, None] - centers[None]) ** 2).sum(-1)
        live = (dist2 < (1.6 * radii[None]) ** 2).any(1).nonzero(as_tuple=True)[0]
        aabb = torch.tensor([[[-1.0, 1.0]] * 3], device=dev)
        for lo in range(0, live.numel(), chunk):
            idx = live[lo:lo + chunk]
            oo, dd = o[idx].contiguous(), d[idx].contiguous()
            near, far, _, hit = F.aabb_intersection_torch(oo, dd, aabb, 1e-7)
            z = near + (far - near) * torch.linspace(0, 1, n, device=dev)[None]
            x = (oo[:, None] + dd[:, None] * z[..., None]).reshape(-1, 3)
            s, c = field(x, dd[:, None].expand(-1, n, -1).reshape(-1, 3))
            s = s.view(-1, n) * hit.float()
            res = F.ray_marching_fwd(s, c.view(-1, n, 3), z.contiguous(), white_bkg=False)
            acc = res['mask'].clamp(0, 1)
            straight = torch.where(acc[:, None] > 1e-6, res['rgb'] / acc[:, None].clamp_min(1e-6), torch.zeros_like(res['rgb'])).clamp(0, 1)
            out[idx] = torch.cat([straight, acc[:, None]], -1)
        return torch.round(out * 255.0).to(torch.uint8)

    t0 = time.perf_counter()
    Kt, Mt = _cameras(0, N_TRAIN, dev)
    Ke, Me = _cameras(1000, N_TEST, dev)
    rgba = torch.stack([render(Kt[v], Mt[v]) for v in range(N_TRAIN)])
    rgba_e = torch.stack([render(Ke[v], Me[v]) for v in range(N_TEST)]).float() / 255.0
    target = rgba_e[..., :3] * rgba_e[..., 3:] + (1.0 - rgba_e[..., 3:])          # blend_bkg_color [1, 1, 1] (the eval augmentation of the yaml)
    torch.cuda.synchronize()
    sc = {'rgba': rgba, 'K': Kt, 'c2w': Mt, 'K_test': Ke, 'c2w_test': Me, 'target': target, 'seconds': time.perf_counter() - t0,
          'covered': float((rgba[..., 3] > 0).float().mean())}
    _SCENE[dev] = sc
    return sc


def run(max_it=2000, seed=0, verbose=False, stepper='fused', report=(100, 500, 2000, 10000, 30000, 50000), window=None):
    """window = (first, last): those iterations of THIS loop (applied occupancy refreshes, fresh shuffled batches against the scene's pixels,
    dynamic batch size) are timed between two device synchronisations and their valid samples counted: -> 'training_loop'"""
    sc = build_image_scene()
    dev = sc['rgba'].device
    from arcnerf_amd.ops.volume_func import sampler_rng
    sampler_rng(reset=True)
    torch.manual_seed(int(seed))
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), EXPR_MODEL)).to(dev)
    fg = m.fg_model
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-1, eps=1e-15, weight_decay=1e-6, ema_decay=0.95,
                    zero_grad_on_step=True, ema_in_param=True).flatten()
    ema = T.EMA(m, 0.95, opt)
    lc = dict_to_obj({'loss': {'ImgLoss': {'keys': ['rgb_coarse'], 'loss_type': 'Huber', 'delta': 0.1, 'weight': 3000.0}}})
    loss_factory = T.build_loss(lc)
    pipe = T.Pipeline()
    pipe.set_n_rays(None, 4096)
    pipe.setup_cfgs(dict_to_obj(SCHEDULER))
    data = {'rgba': sc['rgba'], 'intrinsic': sc['K'], 'c2w': sc['c2w'], 'H': HW, 'W': HW}
    batches = T.TrainBatches(pipe, lambda: data)
    step = T.FusedNgpStep(m, loss_factory, opt, ema, total_epoch=max_it) if stepper == 'fused' else None
    test_rays = [get_rays(HW, HW, sc['K_test'][v], sc['c2w_test'][v], wh_order=False, center_pixel=True) for v in range(N_TEST)]

    @torch.no_grad()
    def psnr():
        m.eval()
        mse = 0.0
        for v, (o, d, _, r) in enumerate(test_rays):
            for lo in range(0, o.shape[0], 32768):
                out = m({'rays_o': o[None, lo:lo + 32768], 'rays_d': d[None, lo:lo + 32768], 'rays_r': r[None, lo:lo + 32768]}, inference_only=True)
                mse += float(((out['rgb'][0] - sc['target'][v, lo:lo + 32768]) ** 2).sum())
        m.train()
        return -10.0 * math.log10(mse / sc['target'].numel())

    m.train()
    base_lr = 1e-1
    points, t_train, samples = [], 0.0, 0
    t_last = time.perf_counter()
    win = None
    for epoch in range(max_it):
        if window is not None and step is not None and epoch == window[0]:
            s_before = step.drain()
            win = {'t0': time.perf_counter(), 'rays0': pipe.get_info('n_rays')}
        opt.param_groups[0]['lr'] = base_lr * (0.33 ** sum(1 for s in (20000, 30000, 40000, 50000) if epoch >= s))      # MultiStepLR of the yaml
        t_call = time.perf_counter()
        T.train_epoch(m, batches, loss_factory, opt, ema, pipe, epoch, total_epoch=max_it, stepper=step)
        if win is not None and 't1' not in win:
            win['host'] = win.get('host', 0.0) + time.perf_counter() - t_call
        if win is not None and 't1' not in win and epoch + 1 == window[1]:
            s_after = step.drain()
            win['t1'] = time.perf_counter()
            n_it = window[1] - window[0]
            every = fg.obj_bound.get_optim_cfgs('epoch_optim')
            win = {'iterations': [window[0], window[1]], 'ms_per_step': (win['t1'] - win['t0']) * 1e3 / n_it, 'host_ms_per_step': win.get('host', 0.0) * 1e3 / n_it, 'samples_per_step': (s_after - s_before) / n_it,
                   'samples_per_s': (s_after - s_before) / (win['t1'] - win['t0']), 'rays_per_step': [win['rays0'], pipe.get_info('n_rays')],
                   'occupancy_refreshes_applied': len([e for e in range(window[0], window[1]) if every and e > 0 and e % every == 0]),
                   'occupied': float(fg.obj_bound.volume.get_voxel_bitfield().float().mean()), 't1': win['t1']}
        if (epoch + 1) in report or epoch + 1 == max_it:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_last
            pt = {'iter': epoch + 1, 'psnr': psnr(), 'train_seconds': t_train, 'occupied': float(fg.obj_bound.volume.get_voxel_bitfield().float().mean()),
                  'rays_per_step': pipe.get_info('n_rays'), 'crop': pipe.crop_max_epoch is not None}
            points.append(pt)
            if verbose:
                print(json.dumps(pt), file=sys.stderr, flush=True)
            t_last = time.perf_counter()
    return {'scene': 'analytic: 6 soft blobs, textured, view dependent; {} training views of {} x {} RGBA bytes, {} held out'.format(N_TRAIN, HW, HW, N_TEST),
            'recipe': 'nerf_lego_nerf_ngp.yaml: precrop 0.5 / 500, random bkg colour, cross-view shuffle, dynamic batch size 16 from 4096 rays, Adam 1e-1 + EMA 0.95',
            'path': 'build_model(nerf_ngp.yaml) + trainer.train_epoch + trainer.TrainBatches + ' + ('trainer.FusedNgpStep' if step is not None else 'trainer.step_optimize'),
            'seed': int(seed), 'data_seconds': sc['seconds'], 'pixels_covered': sc['covered'], 'points': points,
            'training_loop': {k: v for k, v in win.items() if k != 't1'} if win is not None and 't1' in win else None,
            'fused_steps': step.steps if step is not None else 0, 'buffer_rebuilds': step.rebuilds if step is not None else 0}


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seeds = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
    mode = os.environ.get('ARCN_MODULE_STEP', 'fused')
    print(json.dumps({'iterations': n, 'runs': [run(n, seed=s, verbose=True, stepper=mode) for s in seeds]}))
