"""BASELINE config 4 AS A COMPOSITE against runs of the reference's FullModel (golden G25, tests/golden/make_golden_composite.py):
  ngpmv_   NeuS on the hash grid in the occupancy-pruned volume + MultiVol cascade background (capture_qqtiger_neusngp_multivol.yaml),
  neuspp_  NeuS 8 x 256 + NeRF++ inverted-sphere background (capture_qqtiger_neus_nerfpp.yaml, the yaml's full widths),
both `bkg_blend: rgb`, built by `build_model` of the mirror from the same yaml, reference state_dict, the reference run's taped uniforms
(`perturb: True` as the yamls have it), samplers K2 / K3 / K11 on the HIP kernels with the pcg32 streams of the run.  Bars: sample
positions and masks bit for bit, outputs 1e-4 (north_star), loss 1e-5, every gradient within 1e-3 of its max."""
import os
import tempfile

import numpy as np
import pytest
import torch

import seeded_weights as SW
from conftest import ROOT, load_golden
from parity_bars import check_against_float64, grad_bar
from rand_feed import RandFeed

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, 'configs')


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def close(a, b, rtol=1e-4, atol=1e-4):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def close_normals(a, b):
    """per-SAMPLE sdf gradients (not an rgb / depth output): d/dx of a 2^9-frequency embedding (or of a hash level of resolution 64+)
    turns an ulp of the position into ~1e-4 of phase, so single components differ by a few 1e-4 between any two fp32 evaluations;
    a MOVED sample would differ by O(1).  Bar: every component within 1e-3, all but 0.1 % within 2e-4."""
    np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-3)
    assert (np.abs(a - b) > 2e-4 + 2e-4 * np.abs(b)).mean() < 1e-3


class Sub:
    """the entries of one composite (a key prefix) of the fixture, with the npz interface the helpers expect"""

    def __init__(self, g, tag):
        self.g, self.tag = g, tag
        self.files = [k[len(tag):] for k in g.files if k.startswith(tag)]

    def __getitem__(self, k):
        return self.g[self.tag + k]


def _loss(out, inputs):
    eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    return ((out['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik, eik


def _check_all_grads(m, g, rtol=1e-3):
    n_full = n_sum = 0
    floor = rtol
    for n, p in m.named_parameters():
        rtol = grad_bar(g, n, floor)        # 1e-3, or 1.25 x the reference's own fp32-vs-float64 error where that is larger (parity_bars.py)
        if p.grad is not None:
            check_against_float64(g, n, p.grad.cpu().numpy(), rtol)
        if ('gsum.' + n + '.max') in g.files:
            SW.check_grad({k: g['gsum.' + n + '.' + k] for k in ('head', 'mod16', 'sum', 'abs', 'max', 'proj')}, p.grad.cpu().numpy(), rtol=rtol, name=n)
            n_sum += 1
        elif ('grad.' + n) in g.files:
            ref = g['grad.' + n]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= rtol * np.abs(ref).max() + 1e-8, n
            n_full += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    return n_full, n_sum


@pytest.mark.parametrize('packed_fg', [False, True])
def test_neus_on_hashgrid_with_multivol_background_matches_reference_composite(gpu, packed_fg):
    """config 4 as the reference names it: Neus(volume bound K2 / K3, hash encoder, normals through the encoder) + MultiVol (K11) blended
    by `rgb += T_fg,last * rgb_bkg` (full_model.py:278-330).  packed_fg: the foreground on its packed path (csrc/neus.hip: no padded
    (rays, P) tensors, K2 + K3 fused) instead of the dense reference-shaped one - same bars against the same reference run."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.models.base_modules.obj_bound import volume_bound as VB
    from arcnerf_amd.ops import functional as Fn
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = Sub(load_golden('g25_composite_models'), 'ngpmv_')
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(str(g['config_yaml']))
    try:
        m = build_model(load_configs(f.name, [])).to(gpu)
    finally:
        os.unlink(f.name)
    fg, bkg = m.fg_model, m.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'MultiVol' and fg.packed_path_eligible()
    fg.use_packed_path = packed_fg
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
    for name, seed in (('fg_model.geo_net.embed_fn.embeddings', 1), ('bkg_model.geo_net.embed_fn.embeddings', 2)):
        shape = dict(m.named_parameters())[name].shape
        sd[name] = (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) - 0.5) * 0.2
        assert abs(float(sd[name].double().sum()) - float(g['tablesum.' + name])) < 1e-6     # the very table of the reference run
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(('.volume_pts', '.grid_pts', '.corner')) for k in missing), (missing, unexpected)
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    assert not [k for k in g.files if k.startswith('draw_')]      # this composite draws no uniforms (marchers jitter with their own pcg32)

    # record what the two samplers hand to the model
    seen = {'k3': [], 'k11': []}
    real_k3, real_k11 = VB.sparse_volume_sampling, Fn.sparse_sampling_in_multivol_bitfield

    def rec_k3(*a, **k):
        z, msk = real_k3(*a, **k)
        seen['k3'].append((z.clone(), msk.clone()))
        return z, msk

    def rec_k11(*a, **k):
        r = real_k11(*a, **k)
        # (the packed path asks for dense=False: only the first counts[r] entries of a row are written, the mask is not filled in)
        msk = r[1].clone() if (r[1] is not None and k.get('dense', True)) else None
        seen['k11'].append((r[0].clone(), msk, r[2].clone() if len(r) > 2 and r[2] is not None else None))
        return r
    VB.sparse_volume_sampling, Fn.sparse_sampling_in_multivol_bitfield = rec_k3, rec_k11
    sampler_rng(reset=True)
    multivol_rng(reset=True)
    try:
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        assert set(out.keys()) == {k[6:] for k in g.files if k.startswith('infer_')}
        for k in out:
            close(out[k].detach().cpu().numpy(), g['infer_' + k])
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    finally:
        VB.sparse_volume_sampling, Fn.sparse_sampling_in_multivol_bitfield = real_k3, real_k11
        sampler_rng(reset=True)
        multivol_rng(reset=True)
    # sample indices bit-exact: both launches of both marchers (inference, training)
    assert len(seen['k3']) == (0 if packed_fg else 2) and len(seen['k11']) == 2
    for c in range(2):
        if not packed_fg:      # (the packed path marches inside one fused launch; its samples are pinned by test_neus_packed_path_* below)
            z, msk = (t.cpu().numpy() for t in seen['k3'][c])
            ref_m = np.unpackbits(g['k3_call{}_mask'.format(c)], axis=1, bitorder='little')[:, :msk.shape[1]].astype(bool)
            assert np.array_equal(msk, ref_m)
            w = g['k3_call{}_zvals'.format(c)].shape[1]
            assert np.array_equal(z[:, :w].view(np.uint32), g['k3_call{}_zvals'.format(c)].view(np.uint32))
        z, msk, cnt = seen['k11'][c]
        z = z.cpu().numpy()
        ref_m = np.unpackbits(g['k11_call{}_mask'.format(c)], axis=1, bitorder='little')[:, :z.shape[1]].astype(bool)
        if msk is not None:
            assert np.array_equal(msk.cpu().numpy(), ref_m)
        if cnt is not None:
            assert np.array_equal(cnt.cpu().numpy().astype(np.int64), ref_m.sum(1))
        w = g['k11_call{}_zvals'.format(c)].shape[1]
        valid = ref_m[:, :w]
        assert np.array_equal(z[:, :w].view(np.uint32)[valid], g['k11_call{}_zvals'.format(c)].view(np.uint32)[valid])
    for k in [k[6:] for k in g.files if k.startswith('train_') and k not in ('train_loss', 'train_eikonal')]:
        if k == 'normal_pts':
            close_normals(out[k].detach().cpu().numpy(), g['train_' + k])
        else:
            close(out[k].detach().cpu().numpy(), g['train_' + k])
    loss, eik = _loss(out, inputs)
    assert abs(float(eik) - float(g['train_eikonal'])) < 2e-5 and abs(float(loss) - float(g['train_loss'])) < 1e-5
    m.zero_grad()
    loss.backward()
    n_full, _ = _check_all_grads(m, g)
    assert n_full == 13 and 'grad.fg_model.geo_net.embed_fn.embeddings' in g.files and 'grad.bkg_model.geo_net.embed_fn.embeddings' in g.files


@pytest.mark.parametrize('packed_fg', [False, True])
def test_neus_on_hashgrid_with_nerfpp_background_matches_reference_composite(gpu, packed_fg):
    """BASELINE.json's wording of config 4, "NeuS-NGP (hashgrid + volume prune) with NeRF++ background": the foreground block of
    capture_qqtiger_neusngp_multivol.yaml + the background block of capture_qqtiger_neus_nerfpp.yaml (nets narrowed), run through the
    reference's FullModel (golden G25 ngppp_): K3 samples bit for bit, outputs 1e-4, loss 1e-5, all 21 gradients 1e-3; the background's
    shell perturbation on the reference's tape; dense and packed foreground."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.models.base_modules.obj_bound import volume_bound as VB
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = Sub(load_golden('g25_composite_models'), 'ngppp_')
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(str(g['config_yaml']))
    try:
        m = build_model(load_configs(f.name, [])).to(gpu)
    finally:
        os.unlink(f.name)
    fg, bkg = m.fg_model, m.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'NeRFPP' and fg.packed_path_eligible()
    fg.use_packed_path = packed_fg
    sd = {k[3:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith('sd.')}
    name = 'fg_model.geo_net.embed_fn.embeddings'
    sd[name] = (torch.rand(dict(m.named_parameters())[name].shape, generator=torch.Generator().manual_seed(3)) - 0.5) * 0.2
    assert abs(float(sd[name].double().sum()) - float(g['tablesum.' + name])) < 1e-6
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(('.volume_pts', '.grid_pts', '.corner')) for k in missing), (missing, unexpected)
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    seen, real_k3 = [], VB.sparse_volume_sampling

    def rec_k3(*a, **k):
        z, msk = real_k3(*a, **k)
        seen.append((z.clone(), msk.clone()))
        return z, msk
    VB.sparse_volume_sampling = rec_k3
    sampler_rng(reset=True)
    draws = [g[k] for k in sorted(k for k in g.files if k.startswith('draw_'))]
    try:
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
        assert set(out.keys()) == {k[6:] for k in g.files if k.startswith('infer_')}
        for k in out:
            close(out[k].detach().cpu().numpy(), g['infer_' + k])
        with RandFeed(draws, gpu):
            out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    finally:
        VB.sparse_volume_sampling = real_k3
        sampler_rng(reset=True)
    assert len(seen) == (0 if packed_fg else 2)
    for c, (z, msk) in enumerate(seen):
        ref_m = np.unpackbits(g['k3_call{}_mask'.format(c)], axis=1, bitorder='little')[:, :msk.shape[1]].astype(bool)
        assert np.array_equal(msk.cpu().numpy(), ref_m)
        w = g['k3_call{}_zvals'.format(c)].shape[1]
        assert np.array_equal(z.cpu().numpy()[:, :w].view(np.uint32), g['k3_call{}_zvals'.format(c)].view(np.uint32))
    for k in [k[6:] for k in g.files if k.startswith('train_') and k not in ('train_loss', 'train_eikonal')]:
        (close_normals if k == 'normal_pts' else close)(out[k].detach().cpu().numpy(), g['train_' + k])
    loss, eik = _loss(out, inputs)
    assert abs(float(eik.detach()) - float(g['train_eikonal'])) < 2e-5 and abs(float(loss.detach()) - float(g['train_loss'])) < 1e-5
    m.zero_grad()
    loss.backward()
    n_full, _ = _check_all_grads(m, g)
    assert n_full == 21 and 'grad.fg_model.geo_net.embed_fn.embeddings' in g.files


def test_neus_with_nerfpp_background_matches_reference_composite(gpu):
    """BASELINE's "NeuS(-NGP) with NeRF++ background": Neus 8 x 256 (sphere bound, four up-sampling rounds, Eikonal through the double
    backward) + NeRFPP (inverted-sphere 4-D inputs, multi-sphere shells), FULL widths of capture_qqtiger_neus_nerfpp.yaml."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = Sub(load_golden('g25_composite_models'), 'neuspp_')
    m = build_model(load_configs(os.path.join(CFG, 'neus_nerfpp.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    fg, bkg = m.fg_model, m.bkg_model
    assert type(fg).__name__ == 'Neus' and type(bkg).__name__ == 'NeRFPP' and fg.geo_net.W == 256 and bkg.coarse_geo_net.W == 256
    m.load_state_dict({k: torch.from_numpy(v) for k, v in SW.state_dict_from_fixture(g).items()})    # strict
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    assert float(g['tie_margin'].min()) >= 2e-6 and fg.get_ray_cfgs('perturb') is True and bkg.get_ray_cfgs('perturb') is True
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {k[6:] for k in g.files if k.startswith('infer_')}
    for k in out:
        close(out[k].detach().cpu().numpy(), g['infer_' + k])
    draws = [g[k] for k in sorted(k for k in g.files if k.startswith('draw_'))]
    assert len(draws) == 6      # coarse depths, four up-sampling rounds, the background's shell radii
    with RandFeed(draws, gpu):
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for k in [k[6:] for k in g.files if k.startswith('train_') and k not in ('train_loss', 'train_eikonal')]:
        if k == 'normal_pts':
            close_normals(out[k].detach().cpu().numpy(), g['train_' + k])
        else:
            close(out[k].detach().cpu().numpy(), g['train_' + k])
    loss, eik = _loss(out, inputs)
    assert abs(float(eik) - float(g['train_eikonal'])) < 1e-5 and abs(float(loss) - float(g['train_loss'])) < 1e-5
    m.zero_grad()
    loss.backward()
    # the reference's own fp32 gradient is further than 1e-3 from its float64 evaluation on exactly these (the Eikonal term through the
    # 2^9-frequency embedding): 4.7e-3 / 4.4e-3
    assert sorted(k[7:] for k in g.files if k.startswith('f64err.') and float(g[k]) > 8e-4) == \
        ['fg_model.geo_net.layers.0.weight_v', 'fg_model.geo_net.layers.5.weight_v']
    n_full, n_sum = _check_all_grads(m, g)
    assert n_full >= 35 and n_sum >= 20


def test_fused_neus_ngp_step_equals_the_module_path(gpu):
    """trainer.FusedNeusNgpStep (config 4 as a hand-ordered kernel chain: no autograd engine, the Eikonal loss on the packed normals, the next
    batch's samplers on a second stream) against the module path on the same batch from the same state: the same losses, the same flat
    gradient - second-order pieces included - and, stepping, the same parameters after three iterations with prefetched samplers."""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs
    n_rays = 1024
    loss_cfg = dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}})
    pool = []
    g = torch.Generator().manual_seed(3)
    for i in range(3):
        o, d = synthetic_rays(n_rays, seed=20 + i, device=gpu, radius=2.2)
        pool.append({'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=gpu),
                     'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(gpu), 'img': torch.rand(1, n_rays, 3, generator=g).to(gpu)})

    def make():
        torch.manual_seed(0)
        m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(gpu)
        m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(gpu), ops='overwrite')
        m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(gpu))
        with torch.no_grad():      # away from the all-zero features of a fresh table: every gradient path carries signal
            m.fg_model.geo_net.embed_fn.embeddings.mul_(200.0)
            m.bkg_model.geo_net.embed_fn.embeddings.mul_(2000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()
        sampler_rng(reset=True)
        multivol_rng(reset=True)
        m.train()
        return m, opt, T.build_loss(loss_cfg)

    # (a) one batch, gradients only
    m, opt, lf = make()
    out = m(dict(pool[0]), inference_only=False, cur_epoch=20000)
    la = lf(pool[0], out)
    opt.zero_grad()
    la['sum'].backward()
    g_a = opt.flat_grads().clone()
    m, opt, lf = make()
    st = T.FusedNeusNgpStep(m, lf, opt)
    st.apply_optimizer = False
    opt.zero_grad()
    out_b, lb = st(pool[0], 20000)
    g_b = opt.flat_grads().clone()
    for k in ('ImgLoss', 'EikonalLoss', 'sum'):
        assert abs(float(la[k]) - float(lb[k])) <= 2e-5 * abs(float(la[k])) + 1e-7, (k, float(la[k]), float(lb[k]))
    for k in ('rgb', 'depth', 'mask', 'normal'):
        assert torch.allclose(out[k], out_b[k], rtol=1e-5, atol=1e-5), k
    scale = float(g_a.abs().max())
    assert scale > 0 and float((g_a - g_b).abs().max()) <= 2e-5 * scale, float((g_a - g_b).abs().max() / scale)
    # every parameter tensor received its gradient (table x 2, sdf net, radiance nets, inv_s)
    for name, p in m.named_parameters():
        if p.requires_grad:
            assert float(p.grad.abs().max()) > 0, name
    # (b) three iterations with the optimiser, the next batch's samplers prefetched: the module path's parameters
    runs = {}
    for mode in ('eager', 'fused', 'fused_two_pass', 'fused_two_ahead', 'fused_geo_chains', 'fused_march_blend', 'fused_one_stream'):
        m, opt, lf = make()
        # (fused_geo_chains: both geometry nets as round 5's chains of dense products instead of arcn_geo2_*; fused_march_blend: the coming
        # batches' samplers queued after the forwards instead of behind the last scatter)
        # fused_one_stream: the background's chains on the foreground's stream instead of beside them on their own
        kw = {'fused_geo': False} if mode == 'fused_geo_chains' else ({'march_at': 'blend'} if mode == 'fused_march_blend' else {})
        if mode == 'fused_one_stream':
            kw = {'bkg_stream': False}
        st = T.FusedNeusNgpStep(m, lf, opt, **kw) if mode != 'eager' else None
        if st is not None:
            assert st.fused_geo == (mode != 'fused_geo_chains') and st.march_at == ('blend' if mode == 'fused_march_blend' else 'opt')
            assert st.bkg_stream == (mode != 'fused_one_stream')
        if mode == 'fused':
            assert st.fuse_adam           # the scatters' chunk owners apply Adam to the table levels they own
        if mode == 'fused_two_pass':
            st.fuse_adam = False          # scatter into .grad, then one optimiser pass over the whole buffer
        losses = []
        for i in range(3):
            if st is not None and mode == 'fused_two_ahead':       # the samplers of the next TWO batches queued ahead
                _, l = st(pool[i], 20000 + i, next_feed_in=pool[i + 1:i + 3] or None)
                assert len(st._ahead) == min(2, 2 - i)
            elif st is not None:
                _, l = st(pool[i], 20000 + i, next_feed_in=pool[i + 1] if i < 2 else None)
            else:
                _, l = T.step_optimize(m, dict(pool[i]), lf, opt, None, 20000 + i)
            losses.append(float(l['sum']))
        torch.cuda.synchronize()
        runs[mode] = (losses, opt.flat_params().clone(), sampler_rng().state, multivol_rng().state)
    # the optimiser inside the scatter is the optimiser after it: the same accumulated gradient (up to the order the records of a row arrive
    # in, which differs from run to run in either form), the same update arithmetic
    lf_, pf = runs['fused'][:2]
    lt_, pt = runs['fused_two_pass'][:2]
    assert max(abs(x - y) / abs(x) for x, y in zip(lf_, lt_)) < 1e-5, (lf_, lt_)
    assert float(((pf - pt).abs() > 1e-3 * float(pf.abs().max())).float().mean()) < 1e-3
    assert runs['fused'][2:] == runs['fused_two_pass'][2:]
    # two batches ahead: the same samples (both generators in the same state), the same trajectory up to the same summation-order noise
    la2, pa2 = runs['fused_two_ahead'][:2]
    assert runs['fused_two_ahead'][2:] == runs['fused'][2:]
    assert max(abs(x - y) / abs(x) for x, y in zip(lf_, la2)) < 1e-5, (lf_, la2)
    assert float(((pf - pa2).abs() > 1e-3 * float(pf.abs().max())).float().mean()) < 1e-3
    for other in ('fused_geo_chains', 'fused_march_blend', 'fused_one_stream'):
        lo_, po = runs[other][:2]
        assert runs[other][2:] == runs['fused'][2:]
        assert max(abs(x - y) / abs(x) for x, y in zip(lf_, lo_)) < 1e-5, (other, lf_, lo_)
        assert float(((pf - po).abs() > 1e-3 * float(pf.abs().max())).float().mean()) < 1e-3
    (la_, pa, ra, rma), (lb_, pb, rb_, rmb) = runs['eager'], runs['fused']
    assert ra == rb_ and rma == rmb
    assert max(abs(x - y) / abs(x) for x, y in zip(la_, lb_)) < 1e-4, (la_, lb_)
    assert float(((pa - pb).abs() > 1e-3 * float(pa.abs().max())).float().mean()) < 1e-3


def test_fused_neus_ngp_step_at_the_bench_size(gpu):
    """BASELINE config 4 at bench.py's size (4096 rays, ~1.25e5 foreground points + ~0.76e5 background samples, both occupancy structures at 5 %):
    the hand-ordered chain's losses, outputs and flat gradient against the module path on the same batch from the same state, and the kept
    corners against the table gathers (ARCN_NEUS_CORNERS): the properties that do not depend on the size, at the size the metric is quoted on"""
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.multivol_func import multivol_rng
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.optim import FusedAdam
    from arcnerf_amd.pipeline import synthetic_bitfield, synthetic_cascade_bits, synthetic_rays
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs
    n_rays = 4096
    loss_cfg = dict_to_obj({'loss': {'ImgLoss': {'loss_type': 'Huber', 'delta': 0.1, 'weight': 5.0}, 'EikonalLoss': {'key': 'normal_pts', 'weight': 0.1}}})
    g = torch.Generator().manual_seed(11)
    o, d = synthetic_rays(n_rays, seed=0, device=gpu, radius=2.2)
    batch = {'rays_o': o.view(1, -1, 3), 'rays_d': d.view(1, -1, 3), 'rays_r': torch.zeros(1, n_rays, 1, device=gpu),
             'bkg_color': torch.rand(1, n_rays, 3, generator=g).to(gpu), 'img': torch.rand(1, n_rays, 3, generator=g).to(gpu)}

    def make():
        torch.manual_seed(0)
        m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml'), [])).to(gpu)
        m.fg_model.obj_bound.volume.update_bitfield(torch.from_numpy(synthetic_bitfield(128, 0.05, seed=0)).to(gpu), ops='overwrite')
        m.bkg_model.density_bitfield.copy_(torch.from_numpy(synthetic_cascade_bits(128, m.bkg_model.n_levels, 0.05, seed=5)).to(gpu))
        with torch.no_grad():
            m.fg_model.geo_net.embed_fn.embeddings.mul_(200.0)
            m.bkg_model.geo_net.embed_fn.embeddings.mul_(2000.0)
        opt = FusedAdam([p for p in m.parameters() if p.requires_grad], lr=5e-4, eps=1e-15).flatten()
        sampler_rng(reset=True)
        multivol_rng(reset=True)
        m.train()
        return m, opt, T.build_loss(loss_cfg)

    m, opt, lf = make()
    out = m(dict(batch), inference_only=False, cur_epoch=20000)
    la = lf(batch, out)
    opt.zero_grad()
    la['sum'].backward()
    g_a = opt.flat_grads().clone()
    grads = {}
    for corners in (True, False):
        m, opt, lf = make()
        st = T.FusedNeusNgpStep(m, lf, opt)
        st.apply_optimizer, st.keep_corners = False, corners
        opt.zero_grad()
        out_b, lb = st(batch, 20000)
        grads[corners] = opt.flat_grads().clone()
        for k in ('ImgLoss', 'EikonalLoss', 'sum'):
            assert abs(float(la[k].detach()) - float(lb[k])) <= 2e-5 * abs(float(la[k].detach())) + 1e-7, (k, float(la[k].detach()), float(lb[k]))
        for k in ('rgb', 'depth', 'mask', 'normal'):
            assert torch.allclose(out[k], out_b[k], rtol=1e-5, atol=1e-5), k
        scale = float(g_a.abs().max())
        assert scale > 0 and float((g_a - grads[corners]).abs().max()) <= 2e-5 * scale, float((g_a - grads[corners]).abs().max() / scale)
    # the corners are the table's rows: every non-table gradient is the same bits; the tables' differ by the order their records arrive in
    assert float((grads[True] - grads[False]).abs().max()) <= 2e-6 * float(g_a.abs().max())
