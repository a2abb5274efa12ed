"""G25: BASELINE config 4 AS A COMPOSITE, from the reference (run in the build container only).

The reference trains two foreground + background composites on the Capture scenes:
  ngpmv_   configs/expr/Capture/qqtiger/capture_qqtiger_neusngp_multivol.yaml - NeuS on the hash grid inside the occupancy-pruned
           volume + the MultiVol cascade as background, `bkg_blend: rgb` (full_model.py:278-330, multivol_bkg_model.py:114-148,204-261,
           neus_model.py:63-104);
  neuspp_  configs/expr/Capture/qqtiger/capture_qqtiger_neus_nerfpp.yaml - NeuS (8 x 256 sdf net, sphere bound) + NeRF++ background
           on inverted-sphere coordinates (nerfpp_bkg_model.py:51-114).
Both are built here with the reference's own `build_model` from the model block of those files (this repo's configs/neus_ngp_multivol.yaml
/ configs/neus_nerfpp.yaml are copies of them) and run through FullModel.forward on CPU: foreground, background, blending,
invalid-ray defaults and autograd are the reference's code.

What cannot run here is replaced exactly as in G16 / G21: the CUDA-only samplers K2 / K3 (volume bound) and K11 (cascade) by this
repo's CPU oracle of those kernels (pcg32 state = the extension's file-static generator at its first launch, advanced 2^32 per launch),
the tcnn back-ends by the reference's torch back-ends; ngpmv_ runs on reduced grids (16^3 volume, 16^3 x 3 cascade, 8-level 2^12 hash
tables), neuspp_ at the FULL widths of the yaml with its big matrices regenerated from a seed (tests/seeded_weights.py, as G22-G24).

Both yamls train with `perturb: True`; the reference's torch.rand draws are taped (tie_probe.RandTape) and fed to the mirror; for
neuspp_ the stored rays are picked from a pool so that every inverse-CDF decision of the up-sampling rounds has a margin (tie_probe.py).
Stored: inputs, state_dict, outputs of an inference pass and of the training pass, loss (rgb MSE + 0.1 Eikonal), every gradient.
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, ROOT)
sys.path.insert(2, os.path.dirname(HERE))
sys.path.insert(3, HERE)
_r = types.ModuleType('pytorch3d.transforms.rotation_conversions')
for _n in ['axis_angle_to_matrix', 'matrix_to_axis_angle', 'matrix_to_rotation_6d', 'rotation_6d_to_matrix']:
    setattr(_r, _n, lambda *a, **k: None)
sys.modules['pytorch3d'] = types.ModuleType('pytorch3d')
sys.modules['pytorch3d.transforms'] = types.ModuleType('pytorch3d.transforms')
sys.modules['pytorch3d.transforms.rotation_conversions'] = _r
import warnings  # noqa: E402

warnings.filterwarnings('ignore')
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import arcnerf.geometry.volume as ref_volume  # noqa: E402
import arcnerf.models.base_modules.obj_bound.volume_bound as ref_vb  # noqa: E402
import arcnerf.models.multivol_bkg_model as ref_mv  # noqa: E402
import arcnerf.models.neus_model as ref_neus  # noqa: E402
import arcnerf.ops.multivol_func as ref_mv_ops  # noqa: E402
import arcnerf.render.ray_helper as ref_rh  # noqa: E402
import seeded_weights as SW  # noqa: E402
from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tie_probe import (ProbeU, RandTape, float64_gradients, inference_flip_sensitivity, inference_lattice_margin,  # noqa: E402
                       store_fp32_error)

_state = {}
_orig_aabb = ref_volume.aabb_ray_intersection


def oracle_k3(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance=0.0):
    h = _state['rng_vol']
    z, m, c = orc.sparse_volume_sampling(rays_o.numpy(), rays_d.numpy(), near.numpy(), far.numpy(), n_pts, np.float32(dt),
                                         aabb_range.permute(1, 0).contiguous().numpy(), n_grid, bitfield.numpy(), near_distance, h.state, h.inc)
    h.advance()
    _state['k3'].append((z.copy(), m.copy()))
    return torch.from_numpy(z), torch.from_numpy(m)


def oracle_k2(rays_o, rays_d, aabb_range, eps=1e-7, force_torch=False):
    if force_torch:
        return _orig_aabb(rays_o, rays_d, aabb_range, eps, True)
    near, far, pts, mask = orc.aabb_intersection(rays_o.numpy(), rays_d.numpy(), aabb_range.permute(0, 2, 1).contiguous().numpy())
    return torch.from_numpy(near), torch.from_numpy(far), torch.from_numpy(pts), torch.from_numpy(mask)


def oracle_k11(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb_range, aabb_range, n_grid, n_cascade,
               bitfield, near_distance=0.0, inclusive=False):
    h = _state['rng_mv']
    z, m, c = orc.sparse_sampling_in_multivol_bitfield(
        rays_o.numpy(), rays_d.numpy(), near.numpy(), far.numpy(), n_pts, cone_angle, min_step, max_step,
        min_aabb_range.permute(1, 0).contiguous().numpy(), aabb_range.permute(1, 0).contiguous().numpy(), n_grid, n_cascade,
        bitfield.numpy(), near_distance, inclusive, h.state, h.inc)
    h.advance()
    _state['k11'].append((z.copy(), m.copy()))
    return torch.from_numpy(z), torch.from_numpy(m)


ref_vb.CUDA_BACKEND_AVAILABLE = True
ref_vb.sparse_volume_sampling = oracle_k3
ref_mv_ops.CUDA_BACKEND_AVAILABLE = True
ref_mv.CUDA_BACKEND_AVAILABLE = True
ref_mv.sparse_sampling_in_multivol_bitfield = oracle_k11


def build_from_text(text, overrides=()):
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(text)
    try:
        return build_model(load_configs(f.name, list(overrides)), None)
    finally:
        os.unlink(f.name)


def loss_of(res, inputs):
    eik = ((res['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    return ((res['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik, eik


def store_run(out, tag, model, inputs, res, seeded=False):
    loss, eik = loss_of(res, inputs)
    model.zero_grad()
    loss.backward()
    out[tag + 'train_loss'], out[tag + 'train_eikonal'] = loss.detach().numpy(), eik.detach().numpy()
    for k, v in res.items():
        if torch.is_tensor(v):
            out[tag + 'train_' + k] = v.detach().numpy()
    n_grad = 0
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        n_grad += 1
        if seeded and (tag + 'seeded_mean.' + k) in out:
            for kk, vv in SW.grad_summary(p.grad.numpy()).items():
                out[tag + 'gsum.' + k + '.' + kk] = vv
        else:
            out[tag + 'grad.' + k] = p.grad.numpy().copy()
    return float(loss), float(eik), n_grad


# ---- ngpmv_: NeuS on the hash grid in the pruned volume + MultiVol background --------------------------------------------------------
def edited_ngpmv():
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml')))
    ref = yaml.safe_load(open('/root/reference/configs/expr/Capture/qqtiger/capture_qqtiger_neusngp_multivol.yaml'))
    assert cfg['model'] == ref['model'], 'configs/neus_ngp_multivol.yaml is no longer the model block of the reference file'
    m = cfg['model']
    small = {'backend': 'torch', 'n_levels': 8, 'hashmap_size': 12, 'base_res': 4, 'max_res': 64}
    m['obj_bound']['volume']['n_grid'] = 16
    m['rays']['n_sample'] = 96
    m['chunk_pts'] = 4096
    m['geometry']['encoder'].update(small)
    m['radiance']['encoder']['view']['backend'] = 'torch'
    b = m['background']
    b['basic_volume'].update({'n_grid': 16, 'n_cascade': 3})
    b['rays'].update({'n_sample': 96, 'cone_angle': 0.0625})
    b['chunk_pts'] = 4096
    b['geometry']['encoder'].update(dict(small, side=6.0))
    b['radiance']['encoder']['view']['backend'] = 'torch'
    return yaml.dump({'model': m}, default_flow_style=False)


def ngpmv(out):
    tag = 'ngpmv_'
    text = edited_ngpmv()
    ref_volume.aabb_ray_intersection = oracle_k2
    torch.manual_seed(2501)
    model = build_model_checked(text)
    fg, bkg = model.fg_model, model.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'MultiVol'
    rng = np.random.default_rng(25)
    with torch.no_grad():   # tables in +-0.1 instead of +-1e-4, a pruned occupancy grid on both sides
        for emb, seed in ((fg.geo_net.embed_fn.embeddings, 1), (bkg.geo_net.embed_fn.embeddings, 2)):
            emb.copy_((torch.rand(emb.shape, generator=torch.Generator().manual_seed(seed)) - 0.5) * 0.2)
        vol_bits = rng.random((16, 16, 16)) < 0.45
        fg.obj_bound.volume.get_voxel_bitfield().copy_(torch.from_numpy(vol_bits))
        cas_bits = rng.random(16 ** 3 * 2) < 0.35
        bkg.density_bitfield.copy_(torch.from_numpy(np.packbits(cas_bits, bitorder='little')))
        for n_, p in bkg.named_parameters():
            if n_.endswith('geo_net.layers.1.weight'):
                p[:1] += 0.25      # some background density everywhere
    g = torch.Generator().manual_seed(2502)
    N = 160
    o = torch.randn(1, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (1.2 + 1.5 * torch.rand(1, N, 1, generator=g))   # cameras between the cascades
    o[:, :24] = (torch.rand(1, 24, 3, generator=g) - 0.5) * 1.2                              # ... and 24 inside the inner volume
    d = -o + (torch.rand(1, N, 3, generator=g) - 0.5) * 1.4
    d[:, 140:] = torch.randn(1, 20, 3, generator=g)                                          # 20 rays looking anywhere (most miss the volume)
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(1, N, 1), 'img': torch.rand(1, N, 3, generator=g),
              'bkg_color': torch.rand(1, N, 3, generator=g)}
    out[tag + 'config_yaml'] = np.array(text)
    for k, v in model.state_dict().items():
        if k.endswith('embed_fn.embeddings'):      # regenerated by the test from the same torch CPU generator; pinned by a checksum
            out[tag + 'tablesum.' + k] = np.array(v.double().sum().item())
        elif not k.endswith(('.volume_pts', '.grid_pts', '.corner')):
            out[tag + 'sd.' + k] = v.numpy()
    for k, v in inputs.items():
        out[tag + 'in_' + k] = v.numpy()
    _state.update(rng_vol=orc.Pcg32(9121), rng_mv=orc.Pcg32(9121), k3=[], k11=[])
    tape = RandTape(2503)
    with tape.record():
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert not tape.draws
    for k, v in res.items():
        out[tag + 'infer_' + k] = v.detach().numpy()
    with tape.record():
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for i, t in enumerate(tape.draws):
        out[tag + 'draw_{:02d}'.format(i)] = t.numpy()
    loss, eik, n_grad = store_run(out, tag, model, inputs, res)
    for name, calls in (('k3', _state['k3']), ('k11', _state['k11'])):
        for c, (z, m) in enumerate(calls):
            out[tag + '{}_call{}_zvals'.format(name, c)] = z[:, :max(2, int(m.sum(1).max()))]
            out[tag + '{}_call{}_mask'.format(name, c)] = np.packbits(m, axis=1, bitorder='little')
    print(tag, 'outputs', {k: tuple(v.shape) for k, v in res.items() if torch.is_tensor(v)})
    print(tag, 'loss', loss, 'eik', eik, 'params with grad', n_grad, 'draws', [tuple(t.shape) for t in tape.draws],
          'fg samples/ray', [float(m.sum(1).mean()) for _, m in _state['k3']], 'rays with fg samples', [int((m.sum(1) > 0).sum()) for _, m in _state['k3']],
          'bkg samples/ray', [float(m.sum(1).mean()) for _, m in _state['k11']], 'mask mean', float(res['mask'].mean()))
    ref_volume.aabb_ray_intersection = _orig_aabb


# ---- ngppp_: NeuS on the hash grid + NeRF++ background (BASELINE config 4 as BASELINE.json words it) ---------------------------------
def edited_ngppp():
    """The reference ships NeuS-NGP with the MultiVol background and plain NeuS with NeRF++; BASELINE.json's config 4 names the cross
    product "NeuS-NGP with NeRF++ background".  Built from the two yamls' own blocks: the foreground block of
    capture_qqtiger_neusngp_multivol.yaml (reduced grids as in ngpmv_) + the `background` block of capture_qqtiger_neus_nerfpp.yaml
    with its nets narrowed (the full-width NeRF++ nets are covered by neuspp_)."""
    fgcfg = yaml.safe_load(edited_ngpmv())['model']
    bk = yaml.safe_load(open('/root/reference/configs/expr/Capture/qqtiger/capture_qqtiger_neus_nerfpp.yaml'))['model']['background']
    bk['geometry'].update({'W': 64, 'D': 4, 'skips': [2], 'W_feat': 64})
    bk['radiance'].update({'W': 32, 'W_feat_in': 64})
    bk['chunk_pts'] = 4096
    fgcfg['background'] = bk
    return yaml.dump({'model': fgcfg}, default_flow_style=False)


def ngppp(out):
    tag = 'ngppp_'
    text = edited_ngppp()
    ref_volume.aabb_ray_intersection = oracle_k2
    torch.manual_seed(2521)
    model = build_model_checked(text)
    fg, bkg = model.fg_model, model.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'NeRFPP' and type(fg.geo_net.embed_fn).__name__ == 'HashGridEmbedder'
    rng = np.random.default_rng(26)
    with torch.no_grad():
        emb = fg.geo_net.embed_fn.embeddings
        emb.copy_((torch.rand(emb.shape, generator=torch.Generator().manual_seed(3)) - 0.5) * 0.2)
        fg.obj_bound.volume.get_voxel_bitfield().copy_(torch.from_numpy(rng.random((16, 16, 16)) < 0.45))
        for n_, p in bkg.named_parameters():
            if n_.endswith('coarse_geo_net.layers.4.weight'):
                p[:1] += 0.35      # background density that matters
    g = torch.Generator().manual_seed(2522)
    N = 128
    o = torch.randn(1, N, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (1.2 + 1.2 * torch.rand(1, N, 1, generator=g))
    d = -o + (torch.rand(1, N, 3, generator=g) - 0.5) * 1.4
    d[:, 112:] = torch.randn(1, 16, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    inputs = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(1, N, 1), 'img': torch.rand(1, N, 3, generator=g),
              'bkg_color': torch.rand(1, N, 3, generator=g)}
    out[tag + 'config_yaml'] = np.array(text)
    for k, v in model.state_dict().items():
        if k.endswith('embed_fn.embeddings'):
            out[tag + 'tablesum.' + k] = np.array(v.double().sum().item())
        elif not k.endswith(('.volume_pts', '.grid_pts', '.corner')):
            out[tag + 'sd.' + k] = v.numpy()
    for k, v in inputs.items():
        out[tag + 'in_' + k] = v.numpy()
    _state.update(rng_vol=orc.Pcg32(9121), rng_mv=orc.Pcg32(9121), k3=[], k11=[])
    tape = RandTape(2523)
    with tape.record():
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert not tape.draws
    for k, v in res.items():
        out[tag + 'infer_' + k] = v.detach().numpy()
    with tape.record():      # the background's shell radii are perturbed (torch.rand); the foreground's marcher jitters with its own pcg32
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for i, t in enumerate(tape.draws):
        out[tag + 'draw_{:02d}'.format(i)] = t.numpy()
    loss, eik, n_grad = store_run(out, tag, model, inputs, res)
    for c, (z, m) in enumerate(_state['k3']):
        out[tag + 'k3_call{}_zvals'.format(c)] = z[:, :max(2, int(m.sum(1).max()))]
        out[tag + 'k3_call{}_mask'.format(c)] = np.packbits(m, axis=1, bitorder='little')
    print(tag, 'loss', loss, 'eik', eik, 'params with grad', n_grad, 'draws', [tuple(t.shape) for t in tape.draws],
          'rays with fg samples', [int((m.sum(1) > 0).sum()) for _, m in _state['k3']], 'mask mean', float(res['mask'].mean()))
    ref_volume.aabb_ray_intersection = _orig_aabb


def build_model_checked(text, overrides=()):
    m = build_from_text(text, overrides)
    assert m.bkg_model is not None
    return m


# ---- neuspp_: NeuS + NeRF++ background at the yaml's full widths --------------------------------------------------------------------
def neuspp(out, pool=400, keep=40, margin=2e-6, pos_noise=5e-6):
    tag = 'neuspp_'
    ref = yaml.safe_load(open('/root/reference/configs/expr/Capture/qqtiger/capture_qqtiger_neus_nerfpp.yaml'))
    mine = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'neus_nerfpp.yaml')))
    assert mine['model'] == ref['model'], 'configs/neus_nerfpp.yaml is no longer the model block of the reference file'
    overrides = ['--model.chunk_pts', '8192', '--model.background.chunk_pts', '8192']
    text = yaml.dump({'model': ref['model']}, default_flow_style=False)
    torch.manual_seed(2511)
    model = build_model_checked(text, overrides)
    fg, bkg = model.fg_model, model.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'NeRFPP' and fg.geo_net.W == 256
    sub = {}
    # big matrices from a seed (same machinery as G22-G24), then the background's density row lifted so that it matters
    import make_golden_fullwidth as FW
    FW.reseed_big_weights(model, sub, 2511)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.startswith('bkg_model') and n_.endswith('geo_net.layers.8.weight'):
                p[:1] += 0.35
                sub['rowpatch.' + n_] = p[:1].numpy().copy()
    for k, v in sub.items():
        out[tag + k] = v
    out[tag + 'overrides'] = np.array(overrides)
    g, pool_in = FW.rays(2512, 1, pool, 3.0, 1.6)
    n_miss = 4
    tang = torch.cross(pool_in['rays_o'][0, -n_miss:], torch.tensor([0.0, 0.0, 1.0]).expand(n_miss, 3), dim=-1)
    pool_in['rays_d'][0, -n_miss:] = tang / tang.norm(dim=-1, keepdim=True)
    assert fg.get_ray_cfgs('perturb') is True and bkg.get_ray_cfgs('perturb') is True
    hit = fg.obj_bound.get_near_far_from_rays({k: v[0] for k, v in pool_in.items()})[2].view(-1)
    n_hit = int(hit.sum())
    tape = RandTape(2513)
    with tape.record(), ProbeU(tape, ref_rh, ref_neus) as probe:
        model({k: v.clone() for k, v in pool_in.items()}, inference_only=False, cur_epoch=20000)
    pool_draws = [t.clone() for t in tape.draws]
    m_pool = np.full(pool, 1.0)
    m_pool[hit.numpy()] = probe.per_ray()[1]
    assert int((~hit[-n_miss:]).sum()) == n_miss
    flip = inference_flip_sensitivity(model, pool_in, ref_rh)     # inference = deterministic lattice: u = 1.0 vs cdf[-1] is a coin flip
    print(tag, 'rays whose inference outputs depend on the u = 1 decision:', int((flip >= 1e-5).sum()), 'of', pool)
    lat = np.full(pool, 1.0)
    noise = np.zeros(pool)
    lat[hit.numpy()], noise[hit.numpy()] = inference_lattice_margin(model, pool_in, ref_rh, ref_neus)
    m_pool = np.minimum(m_pool, lat)      # one margin for both passes: taped uniforms (training) and lattice points (inference)
    noise[hit.numpy()] = np.maximum(noise[hit.numpy()], probe.per_ray_noise())   # worst-conditioned sample of either pass (tie_probe.position_noise)
    ok = np.nonzero((m_pool >= margin) & hit.numpy() & (flip < 1e-5) & (noise <= pos_noise))[0]
    print('rays with a sample in an ill-conditioned bin (position noise >', pos_noise, '):', int((noise > pos_noise).sum()), 'of', pool)
    sel = np.sort(np.concatenate([ok[:keep - n_miss], np.arange(pool - n_miss, pool)]))
    assert len(sel) == keep
    out[tag + 'flip_sensitivity'], out[tag + 'position_noise'] = flip[sel], noise[sel]
    rows = (np.cumsum(hit.numpy()) - 1)[sel][hit.numpy()[sel]]
    inputs = {k: v[:, sel].contiguous() for k, v in pool_in.items()}

    def pick(t):   # per-ray draws follow the selection; draws shared by all rays (the background's shell radii) stay
        if t.shape[0] == pool:
            return t[sel]
        if t.shape[0] == n_hit and n_hit != pool:
            return t[rows]
        return t
    draws = [pick(t) for t in pool_draws]
    out[tag + 'pool_size'], out[tag + 'pool_sel'], out[tag + 'tie_margin'] = np.array(pool), sel, m_pool[sel]
    print(tag, 'pool', pool, 'hit', n_hit, 'below margin', int((m_pool < margin).sum()), 'min margin kept', m_pool[sel].min(),
          'pool draws', [tuple(t.shape) for t in pool_draws])
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out[tag + 'infer_' + k] = v.detach().numpy()
    with tape.replay(draws), ProbeU(tape, ref_rh, ref_neus) as probe:
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    assert probe.per_ray()[1].min() >= margin
    for i, t in enumerate(draws):
        out[tag + 'draw_{:02d}'.format(i)] = t.numpy()
    for k, v in inputs.items():
        out[tag + 'in_' + k] = v.numpy()
    loss, eik, n_grad = store_run(out, tag, model, inputs, res, seeded=True)
    g32 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    g64, o64, l64 = float64_gradients(model, inputs, draws, loss_of, cur_epoch=20000)
    worst = store_fp32_error(out, tag, g32, g64, o64, res)
    print(tag, 'reference fp32 gradient vs its float64 evaluation (relative to max): worst', sorted(((round(v, 5), k) for k, v in worst.items()), reverse=True)[:5],
          'loss', loss, l64)
    print(tag, 'outputs', {k: tuple(v.shape) for k, v in res.items() if torch.is_tensor(v)})
    print(tag, 'loss', loss, 'eik', eik, 'params with grad', n_grad, 'draws', [tuple(t.shape) for t in draws], 'mask mean', float(res['mask'].mean()))


if __name__ == '__main__':
    out = {}
    ngpmv(out)
    ngppp(out)
    neuspp(out)
    path = os.path.join(HERE, 'g25_composite_models.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')
