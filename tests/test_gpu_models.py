"""GPU tests of the model-level mirror: vanilla NeRF against the reference's own FullModel outputs (G9), the packed NGP
fast path against the dense reference-shaped path, training through torch optimisers, and the compat shims."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, 'configs')


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def close(a, b, rtol=1e-4, atol=1e-4):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_vanilla_nerf_matches_reference_fullmodel(gpu):
    """configs/models/nerf.yaml (reduced widths): reference state_dict loaded into the mirror, same rays -> rgb/depth/mask
    within 1e-4 in inference mode and in train mode (perturb/noise off), parameter gradients within 1e-3 of their max."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g9_nerf_model')
    m = build_model(load_configs(os.path.join(CFG, 'nerf.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask'}
    for k in ('rgb', 'depth', 'mask'):
        assert out[k].shape == g['infer_' + k].shape
        close(out[k].cpu().numpy(), g['infer_' + k])
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    assert set(out.keys()) == {'rgb_coarse', 'depth_coarse', 'mask_coarse', 'rgb_fine', 'depth_fine', 'mask_fine'}
    for k in out:
        close(out[k].detach().cpu().numpy(), g['train_' + k])
    loss = ((out['rgb_fine'] - inputs['img']) ** 2).mean() + ((out['rgb_coarse'] - inputs['img']) ** 2).mean()
    loss.backward()
    for n, p in m.named_parameters():
        ref = g['grad.' + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n


def _ngp_model(gpu, extra=()):
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    from arcnerf_amd.ops.volume_func import sampler_rng
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '256', '--model.geometry.encoder.hashmap_size', '14',
          '--model.geometry.encoder.n_levels', '8', '--model.geometry.encoder.max_res', '256'] + list(extra)
    torch.manual_seed(3)
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)).to(gpu)
    with torch.no_grad():
        m.fg_model.coarse_geo_net.embed_fn.embeddings.mul_(3000.0)
    from arcnerf_amd.pipeline import synthetic_bitfield
    bf = torch.from_numpy(synthetic_bitfield(32, 0.1, seed=4)).to(gpu)
    m.fg_model.obj_bound.volume.update_bitfield(bf, ops='overwrite')
    sampler_rng(reset=True)
    return m


def _rays(gpu, B=2, N=300):
    from arcnerf_amd.pipeline import synthetic_rays
    o, d = synthetic_rays(B * N, seed=21, device=gpu)
    g = torch.Generator().manual_seed(5)
    return {'rays_o': o.view(B, N, 3), 'rays_d': d.view(B, N, 3), 'rays_r': torch.zeros(B, N, 1, device=gpu),
            'img': torch.rand(B, N, 3, generator=g).to(gpu), 'bkg_color': torch.rand(B, N, 3, generator=g).to(gpu)}


def test_neus_matches_reference_fullmodel(gpu):
    """configs/models/neus.yaml (reduced widths): reference state_dict loaded into the mirror, same rays ->
    rgb / depth / mask / normal within 1e-4 in inference mode (sphere bound, 4 rounds of sdf up-sampling, sdf_to_alpha,
    alpha compositing) and in train mode; gradients of rgb-MSE + 0.1 Eikonal - which need the SECOND derivative of the geometry
    net through the normals - within 1e-2 of their max (1e-4 for everything except the two matrices that multiply the
    high-frequency position embedding, which feel the handful of up-sampled positions that differ), inv_s included."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g13_neus_model')
    m = build_model(load_configs(os.path.join(CFG, 'neus.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask', 'normal'}
    for k in ('rgb', 'depth', 'mask', 'normal'):
        assert out[k].shape == g['infer_' + k].shape
        close(out[k].detach().cpu().numpy(), g['infer_' + k], rtol=2e-4, atol=2e-4)
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for k in ('rgb', 'depth', 'mask', 'normal'):
        close(out[k].detach().cpu().numpy(), g['train_' + k], rtol=2e-4, atol=2e-4)
    # per-sample normals: a handful of up-sampled positions fall on the other side of a near-tie in the inverse CDF
    bad = np.abs(out['normal_pts'].detach().cpu().numpy() - g['train_normal_pts']) > 2e-4 + 2e-4 * np.abs(g['train_normal_pts'])
    assert bad.mean() < 1e-3, bad.mean()
    prm = out['params'][0] if isinstance(out['params'], list) else out['params']
    assert abs(prm['scale'] - float(g['train_scale'])) < 1e-3
    eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((out['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    assert abs(float(eik) - float(g['train_eikonal'])) < 1e-5 and abs(float(loss) - float(g['train_loss'])) < 1e-5
    loss.backward()
    checked = 0
    for n, p in m.named_parameters():
        if 'grad.' + n not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        ref = g['grad.' + n]
        embed_fed = n.endswith(('geo_net.layers.0.weight_v', 'geo_net.layers.5.weight_v'))
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= (1e-2 if embed_fed else 1e-4) * np.abs(ref).max() + 1e-7, n
        checked += 1
    assert checked > 20 and 'grad.fg_model.inv_s' in g.files


def test_hdrnerf_matches_reference_fullmodel(gpu):
    """configs/models/hdrnerf.yaml (reduced widths): per-ray exposure times through FullModel, tone-mapping
    MLPs, LDR + HDR compositing passes, unit-exposure output; reference state_dict loaded with strict=True."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g14_hdrnerf_model')
    m = build_model(load_configs(os.path.join(CFG, 'hdrnerf.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'hdr', 'depth', 'mask'}
    for k in out:
        assert out[k].shape == g['infer_' + k].shape
        close(out[k].cpu().numpy(), g['infer_' + k], rtol=2e-4, atol=2e-4)
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    keys = {a + '_' + b for a in ('rgb', 'hdr', 'depth', 'mask', 'unit_exp') for b in ('coarse', 'fine')}
    assert set(out.keys()) == keys
    for k in keys:
        assert out[k].shape == g['train_' + k].shape, k
        close(out[k].detach().cpu().numpy(), g['train_' + k], rtol=2e-4, atol=2e-4)
    unit = sum(((out['unit_exp_' + s] - 0.5) ** 2).mean() for s in ('coarse', 'fine'))
    loss = ((out['rgb_fine'] - inputs['img']) ** 2).mean() + ((out['rgb_coarse'] - inputs['img']) ** 2).mean() + 0.5 * unit
    assert abs(float(loss) - float(g['train_loss'])) < 1e-5
    loss.backward()
    checked = 0
    for n, p in m.named_parameters():
        if 'grad.' + n in g.files:
            ref = g['grad.' + n]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n
            checked += 1
    assert checked > 30


@pytest.mark.parametrize('blend', ['rgb', 'sigma'])
def test_nerfpp_background_matches_reference_fullmodel(gpu, blend):
    """configs/models/nerfpp.yaml (reduced widths, SURVEY.md 8f rank 2, NeRF++ half): NeRF foreground + inverted-sphere
    background on multi-sphere shells, blended in `rgb` mode (T_fg,last * background) and in `sigma` mode (joint compositing of
    both models' samples); reference state_dict with strict=True; outputs within 2e-4, rgb-mode gradients within 1e-3 of max."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g15_nerfpp_model')
    ov = [str(v) for v in g['overrides']] + ([str(v) for v in g['sigma_overrides']] if blend == 'sigma' else [])
    m = build_model(load_configs(os.path.join(CFG, 'nerfpp.yaml'), ov)).to(gpu)
    assert type(m.bkg_model).__name__ == 'NeRFPP' and m.bkg_blend == blend
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    tag = blend + '_'
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask'}
    for k in out:
        close(out[k].cpu().numpy(), g[tag + 'infer_' + k], rtol=2e-4, atol=2e-4)
    for mdl in (m.fg_model, m.bkg_model):
        mdl.set_ray_cfgs('perturb', False)
        mdl.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    assert set(out.keys()) == {'rgb_coarse', 'depth_coarse', 'mask_coarse', 'rgb_fine', 'depth_fine', 'mask_fine'}
    for k in out:
        close(out[k].detach().cpu().numpy(), g[tag + 'train_' + k], rtol=2e-4, atol=2e-4)
    if blend == 'rgb':
        loss = ((out['rgb_fine'] - inputs['img']) ** 2).mean() + ((out['rgb_coarse'] - inputs['img']) ** 2).mean()
        assert abs(float(loss) - float(g['rgb_train_loss'])) < 1e-5
        loss.backward()
        checked = 0
        for n, p in m.named_parameters():
            if 'rgb_grad.' + n in g.files:
                ref = g['rgb_grad.' + n]
                assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n
                checked += 1
        assert checked > 60


def test_ngp_packed_path_equals_dense_reference_shaped_path(gpu):
    from arcnerf_amd.ops.volume_func import sampler_rng
    m = _ngp_model(gpu, ['--model.rays.noise_std', '0.0'])
    fg = m.fg_model
    inputs = _rays(gpu)
    outs, grads = {}, {}
    for packed in (True, False):
        fg.use_packed_path = packed
        sampler_rng(reset=True)  # same jitter stream for both paths
        m.zero_grad()
        o_inf = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        o_tr = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
        assert set(o_inf.keys()) == {'rgb', 'depth', 'mask'}
        assert set(o_tr.keys()) == {'rgb_coarse', 'depth_coarse', 'mask_coarse'}
        assert o_tr['rgb_coarse'].shape == (2, 300, 3) and o_tr['depth_coarse'].shape == (2, 300)
        ((o_tr['rgb_coarse'] - inputs['img']) ** 2).mean().backward()
        outs[packed] = {**{k: v.detach().cpu().numpy() for k, v in o_inf.items()}, **{k: v.detach().cpu().numpy() for k, v in o_tr.items()}}
        grads[packed] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
    for k in outs[True]:
        close(outs[True][k], outs[False][k], rtol=1e-5, atol=1e-5)
    assert set(grads[True]) == set(grads[False]) and len(grads[True]) == 3
    for n in grads[True]:
        ref = grads[False][n]
        assert np.abs(grads[True][n] - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-9, n
    # rays that miss every occupied voxel take the defaults: rgb = bkg colour, depth = depth_far, mask = 0
    d = outs[True]['depth']
    miss = d == 10.0
    assert miss.any() and (outs[True]['mask'][miss] == 0).all()
    np.testing.assert_array_equal(outs[True]['rgb'][miss], inputs['bkg_color'].cpu().numpy()[miss])


def test_packed_path_follows_in_place_changes_of_the_occupancy_grid(gpu):
    """The packed path packs the volume's bool grid to bits only when it has changed (the tensor's version counter): every in-place
    writer - update_bitfield and / or / overwrite, reset_voxel_bitfield, load_state_dict - must be seen by the next forward, and with
    FusedAdam.flatten() the gradients land in the flat buffer directly (no AccumulateGrad pass), equal to the unflattened model's."""
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    m = _ngp_model(gpu, ['--model.rays.noise_std', '0.0'])
    fg, vol = m.fg_model, m.fg_model.obj_bound.volume
    inputs = _rays(gpu)
    g = torch.Generator().manual_seed(3)
    grids = [torch.rand(vol.bitfield.shape, generator=g).to(gpu) < p_ for p_ in (0.3, 0.05)]

    def rewind():      # the same jitter stream for every pass (the pipeline keeps the generator object it was built with)
        r = sampler_rng(reset=True)
        if fg._pipe is not None:
            fg._pipe.rng = r

    def render():
        rewind()
        return m({k: v.clone() for k, v in inputs.items()}, inference_only=True)['rgb'].clone()

    def expect(grid):      # a fresh packing of the same grid
        rewind()
        fg._bits_key = None
        fg._pipe.set_bitfield(grid.reshape(-1))
        fg._bits_key = (id(fg._pipe), vol.bitfield.data_ptr(), vol.bitfield._version)
        return m({k: v.clone() for k, v in inputs.items()}, inference_only=True)['rgb'].clone()

    vol.update_bitfield(grids[0], ops='overwrite')
    a = render()
    key = fg._bits_key
    assert torch.equal(render(), a) and fg._bits_key == key           # unchanged grid: not packed again
    vol.update_bitfield(grids[1], ops='and')
    b = render()
    assert fg._bits_key != key and not torch.equal(a, b) and torch.equal(b, expect(grids[0] & grids[1]))
    vol.update_bitfield(grids[1], ops='or')
    assert torch.equal(render(), expect(grids[1]))
    vol.reset_voxel_bitfield(True)
    c = render()
    assert torch.equal(c, expect(torch.ones_like(grids[0]))) and not torch.equal(c, b)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    vol.update_bitfield(grids[0], ops='overwrite')
    d = render()
    m.load_state_dict(sd)
    assert torch.equal(render(), c) and not torch.equal(d, c)

    # gradients straight into FusedAdam's flat buffer == the unflattened model's .grad
    vol.update_bitfield(grids[0], ops='overwrite')
    grads = {}
    for flat in (False, True):
        if flat:
            opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3).flatten()
            assert all(getattr(p, '_arcn_direct_grad', False) for p in m.parameters() if p.requires_grad)
        m.zero_grad() if not flat else opt.zero_grad()
        rewind()
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
        ((out['rgb_coarse'] - inputs['img']) ** 2).mean().backward()
        grads[flat] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        if flat:     # twice: the second backward ADDS to the flat buffer like AccumulateGrad would
            rewind()
            out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
            ((out['rgb_coarse'] - inputs['img']) ** 2).mean().backward()
            twice = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    assert set(grads[True]) == set(grads[False]) and len(grads[True]) == 3
    for n in grads[True]:
        ref = grads[False][n]
        assert float((grads[True][n] - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-12, n
        assert float((twice[n] - 2 * ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-12, n


def test_packed_path_grows_instead_of_dropping_samples(gpu):
    """An all-occupied grid and a large chunk ask for more samples than the packed buffers hold (1.6 M > 2^20): the path must notice,
    grow and repeat the launch - the reference's dense tensors never drop a sample.  Same outputs as the dense path on every ray."""
    from arcnerf_amd.ops.volume_func import sampler_rng
    m = _ngp_model(gpu, ['--model.rays.noise_std', '0.0'])
    fg = m.fg_model
    fg.obj_bound.volume.update_bitfield(torch.ones(32, 32, 32, dtype=torch.bool, device=gpu), ops='overwrite')
    inputs = _rays(gpu, 1, 12000)
    outs = {}
    for packed in (True, False):
        fg.use_packed_path = packed
        sampler_rng(reset=True)
        with torch.no_grad():
            o = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        outs[packed] = {k: v.cpu().numpy() for k, v in o.items()}
    assert fg._pipe.cap > (1 << 20) and int(fg._pipe.n_dev.item()) > (1 << 20)
    for k in outs[True]:
        close(outs[True][k], outs[False][k], rtol=1e-5, atol=1e-5)
    # ... and a training step through the grown buffers still differentiates
    fg.use_packed_path = True
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    ((out['rgb_coarse'] - inputs['img']) ** 2).mean().backward()
    assert float(fg.coarse_geo_net.embed_fn.embeddings.grad.abs().max()) > 0
    # TRAINING on a fresh model (the first steps of a run: all-ones bitfield, R * n_sample samples): NO sample is dropped in any step -
    # the first step has no history and takes the exact path (one host read), the following ones size the buffers from the last step's
    # samples per ray (read back a step late, no stall) - and every step equals the dense path (fg_model.py:252-262), gradients included
    import warnings
    m2 = _ngp_model(gpu, ['--model.rays.noise_std', '0.0'])
    fg2 = m2.fg_model
    fg2.obj_bound.volume.update_bitfield(torch.ones(32, 32, 32, dtype=torch.bool, device=gpu), ops='overwrite')
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        steps = {}
        for packed in (True, False):
            fg2.use_packed_path = packed
            sampler_rng(reset=True)
            for step in range(2):
                m2.zero_grad()
                n = 12000 if step == 0 else 6000           # the ray count changes between steps (dynamic batch size)
                sub = {k: v[:, :n].contiguous() for k, v in inputs.items()}
                o2 = m2({k: v.clone() for k, v in sub.items()}, inference_only=False)
                ((o2['rgb_coarse'] - sub['img']) ** 2).mean().backward()
                steps[(packed, step)] = ({k: v.detach().cpu().numpy() for k, v in o2.items()},
                                         fg2.coarse_geo_net.embed_fn.embeddings.grad.detach().cpu().numpy().copy())
                if packed and step == 0:
                    assert fg2._pipe.cap > (1 << 20) and int(fg2._pipe.n_dev.item()) > (1 << 20)    # grown within the step itself
        m2.eval()     # flushes the deferred check of the last training step
        assert not [w for w in caught if 'packed NGP path' in str(w.message)]
    for step in range(2):
        for k in steps[(True, step)][0]:
            close(steps[(True, step)][0][k], steps[(False, step)][0][k], rtol=1e-5, atol=1e-5)
        ref = steps[(False, step)][1]
        assert np.abs(steps[(True, step)][1] - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-9


def test_ngp_model_trains_with_torch_adam_and_prunes(gpu):
    m = _ngp_model(gpu)
    fg = m.fg_model
    inputs = _rays(gpu, 1, 2048)
    with torch.no_grad():
        fg.coarse_geo_net.embed_fn.embeddings.div_(3000.0)
    inputs['bkg_color'] = torch.zeros_like(inputs['bkg_color'])
    with torch.no_grad():
        hit = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)['depth'] < 10.0
    tgt = hit[..., None].float() * torch.tensor([0.2, 0.7, 0.4], device=gpu)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-15)
    losses = []
    for it in range(120):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
        loss = ((out['rgb_coarse'] - tgt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.2 * losses[0], (losses[0], losses[-1])
    f = fg.get_dynamicbs_factor()
    assert f > 0
    before = int(fg.obj_bound.volume.get_n_occupied_voxel())
    m.optimize(cur_epoch=16)   # warm-up branch: every cell
    m.optimize(cur_epoch=512)  # steady-state branch: 1/4 random + 1/4 occupied cells
    after = int(fg.obj_bound.volume.get_n_occupied_voxel())
    assert 0 < after <= 32 ** 3 and after != before


def test_compat_shims_run_reference_shaped_calls(gpu):
    import arcnerf_amd.compat as compat
    compat.install()
    import _volume_func
    import tinycudann as tcnn
    o, d = _rays(gpu, 1, 64)['rays_o'][0].contiguous(), _rays(gpu, 1, 64)['rays_d'][0].contiguous()
    aabb = torch.tensor([[[-1.0, -1, -1], [1, 1, 1]]], device=gpu)
    near, far = torch.zeros(64, 1, device=gpu), torch.zeros(64, 1, device=gpu)
    pts, mask = torch.zeros(64, 1, 2, 3, device=gpu), torch.zeros(64, 1, dtype=torch.bool, device=gpu)
    _volume_func.aabb_intersection(o, d, aabb, near, far, pts, mask)
    assert mask.any() and (far[mask] > near[mask]).all()
    with pytest.raises(RuntimeError):
        _volume_func.aabb_intersection(o.cpu(), d, aabb, near, far, pts, mask)
    enc = tcnn.Encoding(3, {'otype': 'HashGrid', 'n_levels': 4, 'n_features_per_level': 2, 'log2_hashmap_size': 10,
                            'base_resolution': 4, 'per_level_scale': 2.0}).to(gpu)
    net = tcnn.Network(8, 16, {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None', 'n_neurons': 64,
                               'n_hidden_layers': 1}).to(gpu)
    x = torch.rand(500, 3, device=gpu)
    y = net(enc(x))
    assert y.shape == (500, 16)
    y.square().mean().backward()
    assert enc.params.grad is not None and net.params.grad is not None and float(net.params.grad.abs().sum()) > 0
    sh = tcnn.Encoding(3, {'otype': 'SphericalHarmonics', 'degree': 4}).to(gpu)
    assert sh((d + 1) / 2).shape == (64, 16)


def _ngp_bitfield_model(gpu, bf_bool, extra=()):
    """same nets as _ngp_model, bounded by a BitfieldBound whose Morton bitfield holds the same occupancy"""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    from arcnerf_amd.ops.bitfield_func import bitfield_rng
    from oracle import oracle as orc
    from test_oracle_bitfield import to_morton_bits
    ov = ['--model.obj_bound.bitfield.n_grid', '32', '--model.rays.n_sample', '256', '--model.geometry.encoder.hashmap_size', '14',
          '--model.geometry.encoder.n_levels', '8', '--model.geometry.encoder.max_res', '256'] + list(extra)
    torch.manual_seed(3)
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp_bitfield.yaml'), ov)).to(gpu)
    with torch.no_grad():
        m.fg_model.coarse_geo_net.embed_fn.embeddings.mul_(3000.0)
        m.fg_model.obj_bound.density_bitfield.copy_(torch.from_numpy(to_morton_bits(bf_bool, orc)))
    bitfield_rng(reset=True)
    return m


def test_ngp_bitfield_bound_equals_volume_bound_and_packed_equals_dense(gpu):
    """BitfieldBound (Morton bits, K5) renders what VolumeBound (bool volume, K3) renders for the same occupancy and weights,
    on the packed fast path and on the dense reference-shaped path, forward and parameter gradients."""
    from arcnerf_amd.ops.bitfield_func import bitfield_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.models.base_modules.obj_bound import BitfieldBound
    mv = _ngp_model(gpu, ['--model.rays.noise_std', '0.0'])
    bf = mv.fg_model.obj_bound.volume.get_voxel_bitfield().cpu().numpy()
    mb = _ngp_bitfield_model(gpu, bf, ['--model.rays.noise_std', '0.0'])
    assert isinstance(mb.fg_model.obj_bound, BitfieldBound) and mb.fg_model.packed_path_eligible()
    mb.load_state_dict({k: v for k, v in mv.state_dict().items() if 'obj_bound' not in k}, strict=False)
    inputs = _rays(gpu)
    res = {}
    for name, m in (('volume', mv), ('bitfield', mb)):
        for packed in (True, False):
            m.fg_model.use_packed_path = packed
            sampler_rng(reset=True)
            bitfield_rng(reset=True)
            m.zero_grad()
            o_inf = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
            o_tr = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
            ((o_tr['rgb_coarse'] - inputs['img']) ** 2).mean().backward()
            res[name, packed] = ({**{k: v.detach().cpu().numpy() for k, v in o_inf.items()},
                                  **{k: v.detach().cpu().numpy() for k, v in o_tr.items()}},
                                 {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None})
    ref_out, ref_grad = res['volume', False]
    assert (ref_out['depth'] < 10.0).any()
    for key in (('volume', True), ('bitfield', True), ('bitfield', False)):
        out, grad = res[key]
        for k in ref_out:
            close(out[k], ref_out[k], rtol=1e-5, atol=1e-5)
        assert set(grad) == set(ref_grad)
        for n in grad:
            assert np.abs(grad[n] - ref_grad[n]).max() <= 1e-3 * np.abs(ref_grad[n]).max() + 1e-9, (key, n)
    # same dense width on both dense paths and bit-identical samples: the two samplers walk the same t lattice
    assert np.array_equal(res['bitfield', False][0]['depth'], res['volume', False][0]['depth'])


def test_ngp_bitfield_model_trains_and_prunes(gpu):
    bf = np.ones((32, 32, 32), bool)
    m = _ngp_bitfield_model(gpu, bf)
    fg = m.fg_model
    inputs = _rays(gpu, 1, 2048)
    with torch.no_grad():
        fg.coarse_geo_net.embed_fn.embeddings.div_(3000.0)
    inputs['bkg_color'] = torch.zeros_like(inputs['bkg_color'])
    o, d = inputs['rays_o'][0], inputs['rays_d'][0]
    t_mid = -(o * d).sum(-1, keepdim=True)
    hit = ((o + t_mid * d).norm(dim=-1) < 0.5)[None]
    tgt = hit[..., None].float() * torch.tensor([0.2, 0.7, 0.4], device=gpu)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-15)
    losses = []
    for it in range(1, 161):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
        loss = ((out['rgb_coarse'] - tgt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if it % 16 == 0:
            m.optimize(cur_epoch=it)   # warm-up refreshes: every cell
    assert losses[-1] < 0.3 * losses[0], (losses[0], losses[-1])
    cnt, ratio = fg.obj_bound.get_bitfield_count()
    assert 0 < ratio < 0.9 and fg.obj_bound.ema_step == 10
    m.optimize(cur_epoch=512)   # steady state: n/4 uniform + n/4 from occupied cells
    assert fg.obj_bound.ema_step == 11
    sd = m.state_dict()
    assert any(k.endswith('obj_bound.density_bitfield') for k in sd) and any(k.endswith('obj_bound.density_grid') for k in sd)


@pytest.mark.parametrize('tag', ['incl_', 'excl_'])
def test_multivol_matches_reference_model_on_oracle_samples(gpu, tag):
    """G16: the reference's MultiVol (configs/models/multivol.yaml, small grids, torch-Linear nets) run on CPU with its CUDA-only
    sampler call replaced by the oracle of that kernel.  Here: the HIP sampler reproduces the stored zvals / masks bit for bit,
    and the mirror model with the reference state_dict (strict) reproduces outputs within 2e-4 and gradients within 1e-3 of max,
    for both `inclusive` settings (rays without samples included)."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g16_multivol_model')
    ov = [str(v) for v in g['overrides']] + ['--model.basic_volume.inclusive', str(tag == 'incl_')]
    m = build_model(load_configs(os.path.join(CFG, 'multivol.yaml'), ov)).to(gpu)
    mv = m.fg_model
    assert type(mv).__name__ == 'MultiVol' and mv.inclusive == (tag == 'incl_')
    m.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + 'sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    multivol_rng(reset=True)
    # the sampler alone, on the two launches the golden run made (inference, then train)
    o, d = inputs['rays_o'][0], inputs['rays_d'][0]
    for call in range(2):
        near, far = mv.get_near_far_from_rays(o, d)
        close(near.cpu().numpy(), g[tag + 'call{}_near'.format(call)], rtol=0, atol=1e-6)
        z, msk = mv.get_zvals_from_near_far(torch.from_numpy(g[tag + 'call{}_near'.format(call)]).to(gpu),
                                            torch.from_numpy(g[tag + 'call{}_far'.format(call)]).to(gpu), mv.get_ray_cfgs('n_sample'), o, d)
        assert np.array_equal(msk.cpu().numpy(), g[tag + 'call{}_mask'.format(call)])
        assert np.array_equal(z.cpu().numpy().view(np.uint32), g[tag + 'call{}_zvals'.format(call)].view(np.uint32))
    assert (g[tag + 'call0_mask'].sum(1) == 0).any()
    multivol_rng(reset=True)
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask'}
    for k in out:
        close(out[k].cpu().numpy(), g[tag + 'infer_' + k], rtol=2e-4, atol=2e-4)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    for k in out:
        close(out[k].detach().cpu().numpy(), g[tag + 'train_' + k], rtol=2e-4, atol=2e-4)
    rgb_key = [k for k in out if k.startswith('rgb')][0]
    loss = ((out[rgb_key] - inputs['img']) ** 2).mean()
    assert abs(float(loss) - float(g[tag + 'train_loss'])) < 1e-5
    loss.backward()
    checked = 0
    for n, p in m.named_parameters():
        if tag + 'grad.' + n in g.files:
            ref = g[tag + 'grad.' + n]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n
            checked += 1
    assert checked >= 8
    multivol_rng(reset=True)


def test_nerf_with_multivol_background_trains_and_prunes(gpu):
    """configs/nerf_multivol.yaml reduced: packed NGP foreground + MultiVol background (hash grid + fused MLPs over the cascade),
    rgb blending; the loss drops, both occupancy structures refresh, and the cascade's refresh equals the oracle's replay."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    from oracle import oracle as orc
    enc = ['hashmap_size', '14', 'n_levels', '8', 'max_res', '256']
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '128', '--model.background.rays.n_sample', '128',
          '--model.background.basic_volume.n_grid', '16', '--model.background.basic_volume.n_cascade', '3',
          '--model.background.basic_volume.side', '2.0', '--model.background.geometry.encoder.side', '8.0',
          '--model.background.rays.cone_angle', '0.03125', '--model.rays.noise_std', '0.0', '--model.background.rays.noise_std', '0.0']
    for pre in ('--model.geometry.encoder.', '--model.background.geometry.encoder.'):
        for k, v in zip(enc[::2], enc[1::2]):
            ov += [pre + k, v]
    torch.manual_seed(7)
    m = build_model(load_configs(os.path.join(CFG, 'nerf_multivol.yaml'), ov)).to(gpu)
    bkg = m.bkg_model
    assert type(bkg).__name__ == 'MultiVol' and not bkg.inclusive and bkg.total_n_elements == 2 * 16 ** 3
    sampler_rng(reset=True)
    multivol_rng(reset=True)
    inputs = _rays(gpu, 1, 1024)
    inputs['bkg_color'] = torch.zeros_like(inputs['bkg_color'])
    tgt = (inputs['rays_d'] * 0.5 + 0.5).clamp(0, 1)   # a sky that depends on the viewing direction: the background has to learn it
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, eps=1e-15)
    losses = []
    for it in range(1, 81):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
        loss = ((out['rgb_coarse'] - tgt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if it % 16 == 0:
            m.optimize(cur_epoch=it)
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert bkg.ema_step == 5 and int(bkg.density_bitfield.min()) < 255
    trained = [p for p in bkg.parameters() if p.requires_grad]
    assert len(trained) == 3   # hash table + two fused weight blocks (the volumes' origin / side are frozen parameters)
    for p in trained:
        assert p.grad is not None and torch.isfinite(p.grad).all()
    # one more refresh, replayed through the oracle from the current grid (steady state: n/4 uniform + n/4 from occupied cells)
    grid = bkg.density_grid.cpu().numpy().copy()
    h, state = orc.Pcg32(9121), multivol_rng()
    while (h.state, h.inc) != (state.state, state.inc):   # bring a host generator to the module's current position
        h.advance()
    seen = {}
    orig = bkg.get_est_opacity

    def spy(dt, pts):
        seen['pts'] = pts.detach().cpu().numpy()
        seen['opa'] = orig(dt, pts)
        return seen['opa']

    bkg.get_est_opacity = spy
    m.optimize(cur_epoch=512)
    bkg.get_est_opacity = orig
    n_q = bkg.total_n_elements // 4
    inner = bkg.basic_volume.get_range().permute(1, 0).contiguous().cpu().numpy()
    pos_u, idx_u = orc.generate_grid_samples_multivol(grid, n_q, inner, 5, 3, 16, -0.01, False, h.state, h.inc)
    h.advance()
    pos_n, idx_n = orc.generate_grid_samples_multivol(grid, n_q, inner, 5, 3, 16, 0.01, False, h.state, h.inc)
    assert np.array_equal(seen['pts'].view(np.uint32), np.concatenate([pos_u, pos_n]).view(np.uint32))
    tmp = np.zeros_like(grid)
    orc.splat_grid_samples(seen['opa'].cpu().numpy(), np.concatenate([idx_u, idx_n]), tmp)
    orc.ema_grid_samples_nerf(tmp, grid, 0.95)
    assert np.array_equal(bkg.density_grid.cpu().numpy().view(np.uint32), grid.view(np.uint32))
    mean = float(bkg.get_density_grid_mean().cpu().numpy()[0])
    assert np.array_equal(bkg.density_bitfield.cpu().numpy(), orc.update_bitfield_multivol(grid, mean, 0.01, 16, 3, False))
    sampler_rng(reset=True)
    multivol_rng(reset=True)


def _model_from_yaml_text(text, gpu, overrides=()):
    import tempfile
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(text)
    try:
        return build_model(load_configs(f.name, list(overrides))).to(gpu)
    finally:
        os.unlink(f.name)


def test_neus_on_hashgrid_matches_reference_fullmodel(gpu):
    """G18: NeuS whose sdf net sits on the hash encoder (the reference's torch backend, differentiated twice by autograd, against
    the kernel graph table-node + xyz-node + arcn_hashgrid_bwd_bwd).  Reference state_dict, strict.  Inference on the deterministic
    lattice; the training pass as the yaml has it (`perturb: True`) on the reference run's taped uniforms (tests/rand_feed.py) with
    rays picked so that no inverse-CDF decision sits on a tie (margins stored in the fixture, tests/golden/tie_probe.py): outputs
    within 1e-4, no exceptions; the gradients of rgb-MSE + 0.1 Eikonal - which reach the TABLE through the normals - within 1e-3 of
    their max."""
    from rand_feed import RandFeed
    g = load_golden('g18_neus_ngp_model')
    m = _model_from_yaml_text(str(g['config_yaml']), gpu)
    assert type(m.fg_model.geo_net.embed_fn).__name__ == 'HashGridEmbedder'
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    assert float(g['tie_margin'].min()) >= 2e-6 and m.fg_model.get_ray_cfgs('perturb') is True
    # ... and so that the reference's own outputs move by < 1e-5 when a ray origin moves by one ulp (a random +-0.1 hash table is a rough
    # field: a third of all rays move by more than 2e-5 and are not in the fixture); the reference's gradients then move by <= 4e-4
    assert float(g['ulp_sensitivity'].max()) < 1e-5 and max(float(g[k]) for k in g.files if k.startswith('ulperr.')) < 5e-4

    def near(a, b):
        close(a, b, rtol=1e-4, atol=1e-4)

    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k in ('rgb', 'depth', 'mask', 'normal'):
        near(out[k].detach().cpu().numpy(), g['infer_' + k])
    draws = [g[k] for k in sorted(k for k in g.files if k.startswith('draw_'))]
    with RandFeed(draws, gpu):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for k in ('rgb', 'depth', 'mask', 'normal'):
        near(out[k].detach().cpu().numpy(), g['train_' + k])
    npts, ref = out['normal_pts'].detach().cpu().numpy(), g['train_normal_pts']     # every sample's sdf gradient: no moved sample
    close(npts, ref, rtol=1e-3, atol=1e-3)
    assert (np.abs(npts - ref) > 2e-4 + 2e-4 * np.abs(ref)).mean() < 2e-3
    eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((out['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    assert abs(float(eik) - float(g['train_eikonal'])) < 2e-5 and abs(float(loss) - float(g['train_loss'])) < 2e-5
    loss.backward()
    checked = 0
    for n, p in m.named_parameters():
        if 'grad.' + n not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        ref = g['grad.' + n]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-8, n
        checked += 1
    assert checked == 7 and 'grad.fg_model.geo_net.embed_fn.embeddings' in g.files


def test_neus_packed_path_equals_dense_reference_shaped_path(gpu):
    """Neus on the occupancy-marched volume: the packed path (K2 + K3 fused, section layout, one render kernel per direction) against the
    dense path (padded (rays, P) tensors, masks, gathers / scatters: the reference's shape) on the same sampler stream - inference and
    training outputs incl. the per-slot `normal_pts` and the last transmittance FullModel blends with, every gradient (hash table through
    the normals, sdf / radiance nets, the variance parameter); rays that miss the volume or find no occupied cell take the defaults."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import load_configs
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '128', '--model.geometry.encoder.n_levels', '8',
          '--model.geometry.encoder.hashmap_size', '13', '--model.geometry.encoder.max_res', '128', '--model.background', 'None']
    torch.manual_seed(7)
    cfgs = load_configs(os.path.join(CFG, 'neus_ngp_multivol.yaml'), ov[:-2])
    del cfgs.model.background
    m = build_model(cfgs).to(gpu)
    fg = m.fg_model
    assert type(fg).__name__ == 'Neus' and fg.packed_path_eligible() and m.bkg_model is None
    fg.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(32, 0.25, seed=3)).to(gpu), ops='overwrite')
    with torch.no_grad():
        fg.geo_net.embed_fn.embeddings.mul_(300.0)
    n = 1500
    o, d = synthetic_rays(n, seed=9, device=gpu, radius=2.2)
    d[-100:] = -d[-100:]                       # rays looking away: no intersection
    g_ = torch.Generator().manual_seed(1)
    inputs = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3).contiguous(), 'rays_r': torch.zeros(1, n, 1, device=gpu),
              'bkg_color': torch.rand(1, n, 3, generator=g_).to(gpu), 'img': torch.rand(1, n, 3, generator=g_).to(gpu)}
    res = {}
    for packed in (True, False):
        fg.use_packed_path = packed
        sampler_rng(reset=True)
        m.zero_grad()
        o_inf = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        o_tr = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
        t_last = fg.forward({k: v.view(-1, v.shape[-1]).clone() for k, v in inputs.items()}, False, 't_last' if packed else True, 20000)['progress_trans_shift'][:, -1]
        loss = ((o_tr['rgb'] - inputs['img']) ** 2).mean() + 0.1 * ((o_tr['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean() + \
            (o_tr['depth'] * 0.01).mean() + (o_tr['normal'] ** 2).mean()
        loss.backward()
        res[packed] = ({**{'i_' + k: v.detach().cpu().numpy() for k, v in o_inf.items()},
                        **{'t_' + k: v.detach().cpu().numpy() for k, v in o_tr.items() if torch.is_tensor(v)}, 't_last': t_last.detach().cpu().numpy()},
                       {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}, o_tr['params'])
    sampler_rng(reset=True)
    assert set(res[True][0]) == set(res[False][0]) and 't_normal_pts' in res[True][0] and res[True][0]['t_normal_pts'].shape[0] == 1
    for k in res[True][0]:
        close(res[True][0][k], res[False][0][k], rtol=1e-5, atol=1e-5)
    scl = [(p_[0] if isinstance(p_, list) else p_)['scale'] for p_ in (res[True][2], res[False][2])]
    assert abs(scl[0] - scl[1]) < 1e-6
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) == 7
    for k in res[False][1]:
        ref = res[False][1][k]
        assert np.abs(res[True][1][k] - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, k
    miss = res[True][0]['t_mask'][0] == 0
    assert miss.sum() >= 100 and (res[True][0]['t_depth'][0][miss] == 10.0).all()
    np.testing.assert_array_equal(res[True][0]['t_rgb'][0][miss], inputs['bkg_color'].cpu().numpy()[0][miss])
    # nothing marched at all: an empty occupancy grid
    fg.use_packed_path = True
    fg.obj_bound.volume.update_bitfield(torch.zeros(32, 32, 32, dtype=torch.bool, device=gpu), ops='overwrite')
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    assert float(out['mask'].abs().max()) == 0.0 and out['normal_pts'].shape[:2] == (1, n)


def test_neus_ngp_with_multivol_background_trains(gpu):
    """BASELINE config 4 family, reduced (configs/neus_ngp_multivol.yaml): NeuS on the hash grid inside the occupancy-pruned
    volume (sparse sampler, masked samples, second-order path) + MultiVol background, rgb blending.  A few optimiser steps with
    the rgb + Eikonal loss: finite gradients on every trained tensor of both models, loss going down, both prunings refresh."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    small = ['n_levels', '8', 'hashmap_size', '13', 'max_res', '128']
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '96', '--model.rays.n_importance', '0',
          '--model.background.rays.n_sample', '96', '--model.background.basic_volume.n_grid', '16',
          '--model.background.basic_volume.n_cascade', '3', '--model.background.geometry.encoder.side', '6.0',
          '--model.background.rays.cone_angle', '0.03125']
    for pre in ('--model.geometry.encoder.', '--model.background.geometry.encoder.'):
        for k, v in zip(small[::2], small[1::2]):
            ov += [pre + k, v]
    torch.manual_seed(11)
    m = build_model(load_configs(os.path.join(CFG, 'neus_ngp_multivol.yaml'), ov)).to(gpu)
    assert type(m.fg_model).__name__ == 'Neus' and type(m.bkg_model).__name__ == 'MultiVol'
    sampler_rng(reset=True)
    multivol_rng(reset=True)
    from arcnerf_amd.pipeline import synthetic_rays
    o, d = synthetic_rays(768, seed=3, device=gpu)
    inputs = {'rays_o': (o * 0.45).view(1, -1, 3).contiguous(), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, 768, 1, device=gpu),
              'bkg_color': torch.zeros(1, 768, 3, device=gpu)}
    tgt = (inputs['rays_d'] * 0.5 + 0.5).clamp(0, 1)
    opt = torch.optim.Adam(m.parameters(), lr=2e-3, eps=1e-15)
    losses = []
    for it in range(1, 41):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=it)
        eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
        loss = ((out['rgb'] - tgt) ** 2).mean() + 0.1 * eik
        opt.zero_grad()
        loss.backward()
        for n, p in m.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
        opt.step()
        losses.append(float(loss))
        if it % 16 == 0:
            m.optimize(cur_epoch=it)
    assert float(m.fg_model.geo_net.embed_fn.embeddings.grad.abs().max()) > 0
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
    assert m.bkg_model.ema_step == 2
    with torch.no_grad():
        res = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(res.keys()) >= {'rgb', 'depth', 'mask', 'normal'} and torch.isfinite(res['rgb']).all()
    sampler_rng(reset=True)
    multivol_rng(reset=True)


def test_direct_gradient_accumulation_equals_autograd_accumulation(gpu):
    """FusedAdam.flatten() marks its parameters so that the hash-grid nodes (first and second order), the fused-MLP node and the packed
    renders ADD their gradients into the flat buffer themselves and return None to autograd.  On the reduced config-4 model (NeuS on the
    hash grid + MultiVol background: every one of those nodes runs) the flat buffer after backward() equals the .grad of the same model
    without an optimiser and of flatten(direct_grads=False); a second backward adds a second copy."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import load_configs
    small = ['n_levels', '8', 'hashmap_size', '13', 'max_res', '128']
    ov = ['--model.obj_bound.volume.n_grid', '32', '--model.rays.n_sample', '96', '--model.rays.n_importance', '0',
          '--model.background.rays.n_sample', '96', '--model.background.basic_volume.n_grid', '16',
          '--model.background.basic_volume.n_cascade', '3', '--model.background.geometry.encoder.side', '6.0',
          '--model.background.rays.cone_angle', '0.03125']
    for pre in ('--model.geometry.encoder.', '--model.background.geometry.encoder.'):
        for k, v in zip(small[::2], small[1::2]):
            ov += [pre + k, v]
    o, d = synthetic_rays(512, seed=3, device=gpu, radius=2.2)
    inputs = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, 512, 1, device=gpu),
              'bkg_color': torch.zeros(1, 512, 3, device=gpu)}
    tgt = (inputs['rays_d'] * 0.5 + 0.5).clamp(0, 1)
    grads = {}
    for mode in ('plain', 'flat', 'direct'):
        torch.manual_seed(11)
        m = build_model(load_configs(os.path.join(CFG, 'neus_ngp_multivol.yaml'), ov)).to(gpu)
        m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(32, 0.3, seed=0)).to(gpu), ops='overwrite')
        with torch.no_grad():
            for e in (m.fg_model.geo_net.embed_fn.embeddings, m.bkg_model.geo_net.embed_fn.embeddings):
                e.mul_(300.0)
        if mode != 'plain':
            opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3).flatten(direct_grads=(mode == 'direct'))
            assert all(getattr(p, '_arcn_direct_grad', None) == (mode == 'direct') for p in m.parameters() if p.requires_grad)
            opt.zero_grad()
        for rep in range(2 if mode == 'direct' else 1):
            sampler_rng(reset=True)
            multivol_rng(reset=True)
            out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
            (((out['rgb'] - tgt) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()).backward()
            if rep == 0:
                grads[mode] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        if mode == 'direct':
            twice = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    sampler_rng(reset=True)
    multivol_rng(reset=True)
    assert set(grads['plain']) == set(grads['flat']) == set(grads['direct']) and len(grads['plain']) >= 8
    for n, ref in grads['plain'].items():
        scale = float(ref.abs().max())
        assert scale > 0, n
        # (the table scatters sum in float: run-to-run order noise ~1e-6 of the largest entry)
        assert float((grads['flat'][n] - ref).abs().max()) <= 1e-4 * scale, n
        assert float((grads['direct'][n] - ref).abs().max()) <= 1e-4 * scale, n
        assert float((twice[n] - 2.0 * ref).abs().max()) <= 2e-4 * scale, n


def test_surface_render_matches_reference(gpu):
    """G19: FullModel.surface_render of the reference on the G13 NeuS model (sphere tracing; secant search on the zero level,
    normals included) and on the G9 NeRF model (secant search on a density level).  The hit masks are identical, depth / rgb /
    normal within 2e-4 (the device-resident loops evaluate frozen rays again instead of gathering the active ones: same results)."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g19_surface_render')

    def load(fixture, cfg):
        f = load_golden(fixture)
        m = build_model(load_configs(os.path.join(CFG, cfg), [str(v) for v in f['overrides']])).to(gpu)
        m.load_state_dict({k[3:]: torch.from_numpy(f[k]) for k in f.files if k.startswith('sd.')})
        return m.eval(), {k[3:]: torch.from_numpy(f[k]).to(gpu) for k in f.files if k.startswith('in_')}

    def check(res, tag, keys):
        assert set(res.keys()) == set(keys)
        assert np.array_equal(res['mask'].cpu().numpy(), g[tag + 'mask'])
        for k in keys:
            assert res[k].shape == g[tag + k].shape, k
            close(res[k].detach().cpu().numpy(), g[tag + k], rtol=2e-4, atol=2e-4)

    neus, inputs = load('g13_neus_model', 'neus.yaml')
    res = neus.surface_render({k: v.clone() for k, v in inputs.items()}, method='sphere_tracing', n_iter=60, threshold=0.002)
    check(res, 'neus_st_', ('rgb', 'depth', 'mask', 'normal'))
    assert 0 < float(res['mask'].sum()) < res['mask'].numel()   # some rays miss the sphere bound: the skipped-ray branch ran
    res = neus.surface_render({k: v.clone() for k, v in inputs.items()}, method='secant_root_finding', n_step=48, n_iter=12, threshold=0.002)
    check(res, 'neus_sec_', ('rgb', 'depth', 'mask', 'normal'))
    with pytest.raises(AssertionError):
        neus.surface_render(inputs, level=1.0)
    nerf, inputs = load('g9_nerf_model', 'nerf.yaml')
    res = nerf.surface_render({k: v.clone() for k, v in inputs.items()}, method='secant_root_finding', n_step=48, n_iter=12,
                              threshold=0.002, level=float(g['nerf_level']), grad_dir='descent')
    check(res, 'nerf_sec_', ('rgb', 'depth', 'mask'))
    with pytest.raises(AssertionError):
        nerf.surface_render(inputs, method='sphere_tracing')
    with pytest.raises(NotImplementedError):
        from arcnerf_amd.geometry.ray import surface_ray_intersection
        surface_ray_intersection(inputs['rays_o'][0], inputs['rays_d'][0], None, method='bisection')


def test_neus_ngp_with_fused_radiance_net_second_order(gpu):
    """the reference's dtu_65_neus_ngp.yaml shape: sdf net = hash encoder + nn.Linear (second order through the encoder), radiance
    = FusedMLPRadianceNet in mode 'pvnf' fed with the NORMALS - the rgb loss reaches the table through the fused net's input
    gradient and the encoder's double backward.  Checked against finite differences of the loss along a random table direction."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    ov = ['--model.obj_bound.volume.n_grid', '16', '--model.rays.n_sample', '48', '--model.rays.n_importance', '0',
          '--model.geometry.encoder.n_levels', '4', '--model.geometry.encoder.hashmap_size', '10', '--model.geometry.encoder.max_res', '32',
          '--model.geometry.encoder.base_res', '4', '--model.radiance.type', 'FusedMLPRadianceNet', '--model.background.type', 'NeRFPP']
    import copy
    cfgs = load_configs(os.path.join(CFG, 'neus_ngp_multivol.yaml'), ov)
    del cfgs.model.__dict__['background']   # foreground only
    torch.manual_seed(5)
    m = build_model(cfgs).to(gpu)
    assert type(m.fg_model.radiance_net).__name__ == 'FusedMLPRadianceNet' and m.fg_model.radiance_net.mode == 'pvnf'
    table = m.fg_model.geo_net.embed_fn.embeddings
    with torch.no_grad():
        table.copy_((torch.rand_like(table) - 0.5) * 0.2)
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.pipeline import synthetic_rays
    o, d = synthetic_rays(256, seed=2, device=gpu)
    inputs = {'rays_o': (o * 0.4).view(1, -1, 3).contiguous(), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, 256, 1, device=gpu),
              'bkg_color': torch.zeros(1, 256, 3, device=gpu)}
    tgt = (inputs['rays_d'] * 0.5 + 0.5).clamp(0, 1)
    m.fg_model.set_ray_cfgs('perturb', False)

    def loss_fn():
        sampler_rng(reset=True)   # same samples every evaluation
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
        return ((out['rgb'] - tgt) ** 2).mean() + 0.1 * ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()

    loss = loss_fn()
    m.zero_grad()
    loss.backward()
    g = table.grad.clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    for p in m.parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all()
    # directional derivative along the (normalised) gradient itself, central differences in float32
    direction = g / g.norm()
    eps = 2e-3
    with torch.no_grad():
        table.add_(eps * direction)
        lp = float(loss_fn())
        table.sub_(2 * eps * direction)
        lm = float(loss_fn())
        table.add_(eps * direction)
    fd, an = (lp - lm) / (2 * eps), float((g * direction).sum())
    assert abs(fd - an) <= 0.05 * abs(an) + 1e-4, (fd, an)
    sampler_rng(reset=True)


@pytest.mark.parametrize('inclusive', [True, False])
def test_multivol_packed_path_equals_dense_path(gpu, inclusive):
    """MultiVol._forward_packed (scan + compaction + packed compositor, no padded tensors) against the dense reference-shaped path
    on the same samples: outputs within 1e-5, parameter gradients within 1e-4 of their max; rays without samples included, and the
    all-empty batch."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g16_multivol_model')
    tag = 'incl_' if inclusive else 'excl_'
    ov = [str(v) for v in g['overrides']] + ['--model.basic_volume.inclusive', str(inclusive)]
    m = build_model(load_configs(os.path.join(CFG, 'multivol.yaml'), ov)).to(gpu)
    m.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + 'sd.')})
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    mv = m.fg_model
    res = {}
    for packed in (True, False):
        mv.use_packed_path = packed
        multivol_rng(reset=True)
        m.zero_grad()
        with torch.no_grad():
            o_inf = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        o_tr = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
        rgb_key = [k for k in o_tr if k.startswith('rgb')][0]
        ((o_tr[rgb_key] - inputs['img']) ** 2).mean().backward()
        res[packed] = ({**{'i_' + k: v.cpu().numpy() for k, v in o_inf.items()}, **{'t_' + k: v.detach().cpu().numpy() for k, v in o_tr.items()}},
                       {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None})
    assert set(res[True][0]) == set(res[False][0])
    for k in res[True][0]:
        close(res[True][0][k], res[False][0][k], rtol=1e-5, atol=1e-5)
    for n in res[False][1]:
        ref = res[False][1][n]
        assert np.abs(res[True][1][n] - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-9, n
    # nothing sampled at all: an empty bitfield
    mv.use_packed_path = True
    with torch.no_grad():
        mv.density_bitfield.zero_()
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert float(out['mask'].abs().max()) == 0.0 and torch.isfinite(out['rgb']).all()
    multivol_rng(reset=True)


def test_linear_radiance_net_fused_route_matches_linear_chain(gpu):
    """RadianceNet (nn.Linear stack, reference state_dict) routes bias-free <= 64-wide ReLU / sigmoid stacks through the fused MLP
    kernel: same parameters, same outputs and gradients (inputs included: the normal's gradient feeds NeuS's second-order path) as the
    layer-by-layer chain; stacks it is not wired for (bias, 256 wide) keep the chain."""
    from arcnerf_amd.models.base_modules.geo_rad_model.linear_network_module import RadianceNet
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    enc = dict_to_obj({'pts': {'input_dim': 3, 'n_freqs': 0, 'type': 'FreqEmbedder'},
                       'view': {'input_dim': 3, 'n_freqs': 0, 'type': 'FreqEmbedder'}})
    torch.manual_seed(0)
    net = RadianceNet(mode='pvnf', W=64, D=2, encoder=enc, W_feat_in=16, use_bias=False).to(gpu)
    assert net._fused_desc is not None
    assert RadianceNet(mode='pvnf', W=64, D=2, encoder=enc, W_feat_in=16, use_bias=True)._fused_desc is None
    assert RadianceNet(mode='pvnf', W=256, D=2, encoder=enc, W_feat_in=16, use_bias=False)._fused_desc is None
    n = 5003
    ins = [torch.randn(n, k, device=gpu, requires_grad=True) for k in (3, 3, 3, 16)]
    tgt = torch.rand(n, 3, device=gpu)
    res = []
    for fused in (True, False):
        desc, net._fused_desc = net._fused_desc, (net._fused_desc if fused else None)
        for p in net.parameters():
            p.grad = None
        out = net(*ins)
        grads = torch.autograd.grad(((out - tgt) ** 2).sum(), ins + [p for p in net.parameters()])
        res.append((out.detach(), [g_.detach() for g_ in grads]))
        net._fused_desc = desc
    assert (res[0][0] - res[1][0]).abs().max() < 1e-5
    for a, b in zip(res[0][1], res[1][1]):
        assert (a - b).abs().max() < 1e-4 * max(1.0, float(b.abs().max()))


def test_extensionless_sampler_semantics_on_gpu(gpu, oracle, monkeypatch):
    """SURVEY a4': what the reference samples WITHOUT its CUDA extension (volume_bound.py:126-141: fixed-step zvals -> occupancy test of
    every sample -> stable compaction), selected by `ops.volume_func.CUDA_BACKEND_AVAILABLE = False`, on GPU tensors with the HIP
    occupancy kernel.  The two helpers reproduce the reference's own outputs (golden G3) on the GPU; the whole chain equals a numpy
    restatement built from the pinned pieces (oracle check_pts_in_occ_voxel = G5)."""
    from conftest import load_golden
    from arcnerf_amd.render.ray_helper import get_zvals_from_near_far_fix_step, handle_valid_mask_zvals
    import arcnerf_amd.ops.volume_func as vf
    g = load_golden('g3_zvals')
    z, m = handle_valid_mask_zvals(torch.from_numpy(g['hv_z']).to(gpu), torch.from_numpy(g['hv_m']).to(gpu))
    assert np.array_equal(m.cpu().numpy(), g['hv_m_out']) and np.array_equal(z.cpu().numpy(), g['hv_z_out'])
    zz, mm = get_zvals_from_near_far_fix_step(torch.tensor([[1.0], [2.0]], device=gpu), torch.tensor([[1.35], [5.0]], device=gpu), 0.1, 6)
    assert mm.cpu().tolist() == [[True, True, True, True, True, False], [True] * 6]
    np.testing.assert_allclose(zz[0].cpu().numpy(), [1.0, 1.1, 1.2, 1.3, 1.35, 1.35], rtol=1e-6)
    # the chain through VolumeBound
    m_ = _ngp_model(gpu)
    bound = m_.fg_model.obj_bound
    from arcnerf_amd.pipeline import synthetic_rays
    o, d = synthetic_rays(500, seed=77, device=gpu)
    near, far, mask_rays = bound.get_near_far_from_rays({'rays_o': o, 'rays_d': d})
    n_pts = 256
    monkeypatch.setattr(vf, 'CUDA_BACKEND_AVAILABLE', False)
    zv, mp = bound.get_zvals_from_near_far(near, far, n_pts, inference_only=True, rays_o=o, rays_d=d)
    monkeypatch.setattr(vf, 'CUDA_BACKEND_AVAILABLE', True)
    # numpy restatement
    vol = bound.volume
    dt = np.float32(vol.get_diag_len() / n_pts)
    nr, fr = near.cpu().numpy(), far.cpu().numpy()
    zs = np.minimum(np.maximum(nr + np.arange(n_pts, dtype=np.float32)[None] * dt, nr), fr).astype(np.float32)
    mk = np.concatenate([np.ones((zs.shape[0], 1), bool), (zs[:, 1:] - zs[:, :-1]) != 0.0], 1)
    on, dn = o.cpu().numpy(), d.cpu().numpy()
    pts = (on[:, None, :] + zs[..., None] * dn[:, None, :]).astype(np.float32)
    aabb23 = vol.get_range().permute(1, 0).contiguous().cpu().numpy()
    occ = oracle.check_pts_in_occ_voxel(pts.reshape(-1, 3), vol.get_voxel_bitfield(flatten=True).cpu().numpy(), aabb23, vol.get_n_grid()).reshape(zs.shape)
    mk &= occ
    ref_z, ref_m = handle_valid_mask_zvals(torch.from_numpy(zs), torch.from_numpy(mk))
    assert np.array_equal(mp.cpu().numpy(), ref_m.numpy())
    assert int(ref_m.sum()) > 1000 and (~ref_m.any(1)).sum() > 10      # rays with and without samples
    # positions computed on the GPU may differ from numpy's by an ulp at voxel faces: compare where the masks agree (everywhere)
    np.testing.assert_allclose(zv.cpu().numpy(), ref_z.numpy(), rtol=1e-6, atol=1e-6)
    # every kept sample lies in an occupied voxel and rows are [T..T F..F] with the tail repeating the last z
    cnt = mp.sum(1)
    cols = torch.arange(n_pts, device=gpu)[None]
    assert torch.equal(mp, cols < cnt[:, None])
