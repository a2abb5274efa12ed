"""The drop-in path end to end: configs/nerf_ngp.yaml -> build_model -> the reference trainer's loop through arcnerf_amd.trainer
(train_epoch: model.optimize -> dynamic batch size -> the step; FusedAdam with the EMA write-back, ImgLoss Huber x 3000) on the analytic
scene of tools/psnr_curve.py (same rays, same ground truth, same seeds) - what a reference user gets after swapping the package.
    ARCN_MODULE_STEP=fused (default): trainer.FusedNgpStep, the module API on the pipeline's fused step
    ARCN_MODULE_STEP=eager:           trainer.step_optimize, every kernel of the module path
usage (GPU box): python tools/train_module_path.py [max_iter=4000]          one run of ARCN_MODULE_STEP, seed PC_SEED
                 python tools/train_module_path.py 2000 0,1,2,3            both steppers for every seed (model init + torch generator), one JSON

The outcome at a given iteration is BIMODAL on this scene with the reference recipe from the module's initialisation (Adam 1e-1, the
occupancy threshold = the mean opacity while nothing has been learned yet): most runs are at 22.5 dB after 500 iterations and 28 - 33 dB
after 2000, some sit at 19.6 dB / 20 - 23 dB - with either stepper (24 processes of seed 0: eager 2 of 12, fused 4 of 12; 14 other seeds:
the same count for both).  Compare distributions, not single runs."""
import importlib.util
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from arcnerf_amd import trainer as T
from arcnerf_amd.models import build_model
from arcnerf_amd.optim import FusedAdam
from arcnerf_amd.utils.cfgs_utils import load_configs


_SCENE = None


def run(max_it=4000, mode=None, seed=0, verbose=True):
    global _SCENE
    mode = mode or os.environ.get('ARCN_MODULE_STEP', 'fused')
    if _SCENE is None:
        spec = importlib.util.spec_from_file_location('psnr_scene', os.path.join(ROOT, 'tools', 'psnr_curve.py'))
        scene = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(scene)
        _SCENE = scene.build_scene()
    sc = _SCENE
    from arcnerf_amd.ops.volume_func import sampler_rng
    sampler_rng(reset=True)
    train, test, dev = sc['train'], sc['test'], sc['dev']
    torch.manual_seed(int(seed))
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), ['--model.rays.white_bkg', 'True'])).to(dev)
    fg = m.fg_model
    # optim block of nerf_lego_nerf_ngp.yaml: Adam 1e-1, eps 1e-15, weight decay 1e-6, WITH the EMA write-back (ema.decay 0.95)
    opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-1, eps=1e-15, weight_decay=1e-6, ema_decay=0.95,
                    zero_grad_on_step=True, ema_in_param=True).flatten()
    ema = T.EMA(m, 0.95, opt)
    lc = type('C', (), {})()
    lc.loss = type('C', (), {})()
    lc.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    loss_factory = T.build_loss(lc)
    tp = T.Pipeline()
    tp.set_info('n_rays', 4096)
    tp.set_info('dynamic_batch_size', 16)                 # dataset.train.scheduler.dynamic_batch_size.update_epoch of the NGP recipe
    tp.set_info('dynamic_max_batch_size', 32768)
    stepper = T.FusedNgpStep(m, loss_factory, opt, ema, total_epoch=max_it + 1) if mode == 'fused' else None
    drawn = [0]

    def get_batch(n_rays):
        drawn[0] += 1
        o, d, tgt, bkg = train[drawn[0] % len(train)]
        return {'rays_o': o[None, :n_rays], 'rays_d': d[None, :n_rays], 'img': tgt[None, :n_rays], 'bkg_color': bkg[None, :n_rays]}

    @torch.no_grad()
    def psnr():
        mse, n = 0.0, 0
        for o, d, tgt in test:
            out = m({'rays_o': o[None], 'rays_d': d[None], 'rays_r': torch.zeros(1, o.shape[0], 1, device=dev),
                     'bkg_color': torch.ones(1, o.shape[0], 3, device=dev)}, inference_only=True)
            mse += float(((out['rgb'][0] - tgt) ** 2).sum())
            n += tgt.numel()
        return -10.0 * math.log10(mse / n)

    m.train()
    points, t_train = [], 0.0
    report = [i for i in (100, 500, 2000, 10000) if i < max_it] + [max_it]
    t_last = time.perf_counter()
    for epoch in range(1, max_it + 1):
        out, loss = T.train_epoch(m, get_batch, loss_factory, opt, ema, tp, epoch, total_epoch=max_it + 1, stepper=stepper)
        if epoch in report:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_last
            m.eval()
            occ = float(fg.obj_bound.volume.get_voxel_bitfield().float().mean())
            points.append({'iter': epoch, 'psnr': psnr(), 'loss': float(loss['sum']), 'train_seconds': t_train, 'occupied': occ,
                           'rays_per_step': tp.get_info('n_rays')})
            m.train()
            if verbose:
                print(json.dumps(points[-1]), file=sys.stderr, flush=True)
            t_last = time.perf_counter()
    return {'path': 'build_model(nerf_ngp.yaml) + trainer.train_epoch + FusedAdam(ema_in_param) + ImgLoss Huber', 'step': mode, 'seed': int(seed),
            'fused_steps': stepper.steps if stepper is not None else 0, 'buffer_rebuilds': stepper.rebuilds if stepper is not None else 0,
            'points': points}


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    if len(sys.argv) > 2:
        runs = [run(n, mode=mode, seed=int(sd), verbose=False) for sd in sys.argv[2].split(',') for mode in ('eager', 'fused')]
        print(json.dumps({'iterations': n, 'runs': runs}))
    else:
        print(json.dumps(run(n, seed=int(os.environ.get('PC_SEED', '0')))))
