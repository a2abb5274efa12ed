"""Run by tests/test_gpu_switches.py in a subprocess with one of the remaining switches at its non-default value (NgpPipeline keyword
switches as a JSON third argument, the library's ARCN_DETERMINISTIC in the environment): small workloads that reach the switched code paths,
results to an .npz for the parent to compare with the default run.
  ngp    : NgpPipeline, 4 training steps with prefetch + occupancy refresh (schedule / scatter switches)
  neusngp: NeuS on the hash grid + the packed NGP module path (first- and second-order table scatters)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, 'configs')
dev = torch.device('cuda:0')


def ngp(out, kw):
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
    cfg = NgpConfig(n_levels=8, hashmap_size=15, max_res=512, n_grid=64, n_sample=512, noise_std=0.0, lr=1e-2)
    fld = NgpField(cfg, device=dev, seed=3)
    fld.view('table').mul_(1000.0)
    pipe = NgpPipeline(fld, max_rays=2048, max_samples=1 << 17, packed_bits=True, **dict({'prefetch_depth': 2}, **kw))
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.1, seed=5)))
    g = torch.Generator().manual_seed(11)
    batches = []
    for i in range(6):
        o, d = synthetic_rays(1024, seed=40 + i, device=dev)
        batches.append((o, d, torch.rand(1024, 3, generator=g).to(dev), torch.rand(1024, 3, generator=g).to(dev)))
    p0 = fld.params.clone()
    for i in range(4):
        o, d, tgt, bkg = batches[i]
        nxt = batches[(i + pipe.prefetch_depth) % 6]
        pipe.train_step(o, d, tgt, bkg_color=bkg, next_rays=(nxt[0], nxt[1]))
        pipe.update_occupancy(16 * (i + 1) + 512, apply=False)
    torch.cuda.synchronize()
    out['ngp_params'] = fld.params.cpu().numpy()
    out['ngp_moved'] = np.array(float((fld.params - p0).abs().max()))
    rgb, depth, mask = pipe.forward(batches[4][0], batches[4][1], batches[4][3], train=False)
    out['ngp_rgb'], out['ngp_depth'] = rgb.cpu().numpy(), depth.cpu().numpy()
    out['ngp_loss'] = np.array(float((rgb - batches[4][2]).abs().mean()))


def _module(name, overrides, n_rays, radius, seed, extra=None):
    from arcnerf_amd.models import build_model
    from arcnerf_amd.pipeline import synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import load_configs
    torch.manual_seed(seed)
    m = build_model(load_configs(os.path.join(CFG, name + '.yaml'), overrides)).to(dev)
    o, d = synthetic_rays(n_rays, seed=seed, device=dev, radius=radius)
    g = torch.Generator().manual_seed(seed)
    inp = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=dev),
           'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(dev), 'img': torch.rand(1, n_rays, 3, generator=g).to(dev)}
    if extra:
        inp.update(extra(g, n_rays))
    return m, inp


def neusngp(out, kw):
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.pipeline import synthetic_bitfield
    small = ['n_levels', '8', 'hashmap_size', '13', 'max_res', '128']
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '96', '--model.background.rays.n_sample', '96',
          '--model.background.basic_volume.n_grid', '16', '--model.background.basic_volume.n_cascade', '3',
          '--model.background.geometry.encoder.side', '6.0', '--model.background.rays.cone_angle', '0.03125']
    for pre in ('--model.geometry.encoder.', '--model.background.geometry.encoder.'):
        for k, v in zip(small[::2], small[1::2]):
            ov += [pre + k, v]
    sampler_rng(reset=True)
    multivol_rng(reset=True)
    m, inp = _module('neus_ngp_multivol', ov, 512, 2.2, 4)
    m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(32, 0.3, seed=0)).to(dev), ops='overwrite')
    with torch.no_grad():
        for e in (m.fg_model.geo_net.embed_fn.embeddings, m.bkg_model.geo_net.embed_fn.embeddings):
            e.mul_(300.0)
    r = m(dict(inp), inference_only=False, cur_epoch=20000)
    (((r['rgb'] - inp['img']) ** 2).mean() + 0.1 * ((r['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()).backward()
    out['neusngp_rgb'] = r['rgb'].detach().cpu().numpy()
    out['neusngp_grad'] = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu().numpy()
    sampler_rng(reset=True)
    m, inp = _module('nerf_ngp', ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '128', '--model.rays.noise_std', '0.0',
                                  '--model.geometry.encoder.n_levels', '8', '--model.geometry.encoder.hashmap_size', '13',
                                  '--model.geometry.encoder.max_res', '128'], 2048, 4.0, 5)
    with torch.no_grad():
        m.fg_model.coarse_geo_net.embed_fn.embeddings.mul_(1000.0)
    r = m(dict(inp), inference_only=False)
    ((r['rgb_coarse'] - inp['img']) ** 2).mean().backward()
    out['ngpmod_rgb'] = r['rgb_coarse'].detach().cpu().numpy()
    out['ngpmod_grad'] = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).cpu().numpy()
    sampler_rng(reset=True)
    multivol_rng(reset=True)


if __name__ == '__main__':
    which, path = sys.argv[1], sys.argv[2]
    kw = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
    out = {}
    {'ngp': ngp, 'neusngp': neusngp}[which](out, kw)
    torch.cuda.synchronize()
    np.savez(path, **out)
