"""Config 2 (instant-ngp) end to end against a RUN OF THE REFERENCE (golden G21, tests/golden/make_golden_ngp.py: the reference's
FullModel -> FgModel.forward -> get_sigma_radiance_by_mask_pts -> HashGridEmbedder -> GeoNet -> RadianceNet -> ray_marching at the
dimensions of configs/models/nerf_ngp.yaml, its CUDA-only K2/K3 calls replaced by the oracle).  The HIP path must reproduce
sample positions and masks bit for bit, rgb / depth / mask within 1e-4 and every gradient within 1e-3 of its max:
  * NgpPipeline (the packed kernels the bench times) for all net / bound variants, including the config's own FUSED nets
    (`nb_fused`: the reference's bias-free nets mapped onto the 32->64->16 / 32->64->64->3 fused shapes);
  * build_model(configs/nerf_ngp.yaml): the dense reference-shaped module path (GeoNet / RadianceNet overrides) and the packed
    module path (the yaml unchanged: FusedMLP types + tcnn back-ends).
"""
import os

import numpy as np
import pytest
import torch

import g21_utils as G
from conftest import ROOT

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, 'configs')


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def g21():
    g = G.golden()
    return g, G.table(g), G.bitfield(g)


def _zvals_packed(g, pre, tag):
    z = g[pre + tag + '_zvals']
    m = G.mask_pts(g, pre + tag + '_mask_pts', z.shape[1])
    return z[m], m.sum(1).astype(np.int32)


@pytest.mark.parametrize('variant,net', [('k2', 'lin'), ('k2', 'nb'), ('k2', 'nb_fused'), ('tb', 'lin')])
def test_pipeline_reproduces_reference_run(gpu, g21, variant, net):
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline
    g, tbl, bf = g21
    cfg = NgpConfig(noise_std=0.0, **G.NETS[net])
    fld = NgpField(cfg, device=gpu, seed=0)
    G.fill_field(fld, g, variant, net, tbl)
    pipe = NgpPipeline(fld, max_rays=256, max_samples=1 << 14, packed_bits=True, torch_aabb=(variant == 'tb'))
    if net == 'nb_fused':
        assert pipe.level_major and pipe.fused_glue     # the kernels of the benchmarked step
    pipe.set_bitfield(torch.from_numpy(bf.reshape(-1)))
    o, d = torch.from_numpy(g['in_rays_o'][0]).to(gpu), torch.from_numpy(g['in_rays_d'][0]).to(gpu)
    bkg, img = torch.from_numpy(g['in_bkg_color'][0]).to(gpu), torch.from_numpy(g['in_img'][0]).to(gpu)
    R = o.shape[0]
    src = 'nb' if net == 'nb_fused' else net
    pre = '{}_{}_'.format(variant, src)

    def check_samples(tag):
        zp, cnt = _zvals_packed(g, pre, tag)
        assert np.array_equal(pipe.buf['counts'][:R].cpu().numpy(), cnt)
        n = int(pipe.n_dev.item())
        assert n == zp.shape[0] and np.array_equal(pipe.buf['t'][:n].cpu().numpy(), zp)      # sample positions: bit-exact
        return cnt

    def outputs(rgb, depth, mask, cnt):
        depth = torch.where(torch.from_numpy(cnt > 0).to(gpu), depth, torch.full_like(depth, 10.0))
        return rgb.cpu().numpy(), depth.cpu().numpy(), mask.cpu().numpy()

    # launch 0 of the sampler's pcg32 stream: inference
    rgb, depth, mask = pipe.forward(o, d, bkg, train=False)
    cnt = check_samples('infer')
    G.check_outputs(g, pre + 'infer_', *outputs(rgb, depth, mask, cnt), train=False)
    # launch 1: train, noise off
    for tag in ('train0', 'train1'):
        noise = None
        if tag == 'train1':
            saved = pipe.rng.state   # launch 2 happens in forward below; the packed noise needs its counts first -> march, rewind
            pipe.sample(o, d)
            cnt2 = pipe.buf['counts'][:R].cpu().numpy()
            pipe.rng._si[0] = saved
            noise = pipe.buf['noise']
            noise.zero_()
            pn = G.packed_noise(g, variant, net, cnt2)
            noise[:pn.shape[0]].copy_(torch.from_numpy(pn))
        rgb, depth, mask = pipe.forward(o, d, bkg, train=True, noise=noise)
        cnt = check_samples(tag)
        G.check_outputs(g, pre + tag + '_', *outputs(rgb, depth, mask, cnt))
        loss = ((rgb - img) ** 2).mean() * 100.0
        assert abs(float(loss) - float(g[pre + tag + '_loss'])) < 1e-3
        d_rgb = (2.0 * (rgb - img) / img.numel() * 100.0).contiguous()
        fld.grads.zero_()
        pipe.backward(o, d, d_rgb)
        torch.cuda.synchronize()
        tg, nets = G.split_flat_grads(fld, fld.grads.cpu().numpy())
        G.check_table_grad(g, pre + tag + '_tgrad_', tg)
        G.check_net_grads(g, variant, net, tag, nets)


def _load_module_weights(m, g, variant, net, tbl, bf, gpu):
    fg = m.fg_model
    w = G.net_weights(g, variant, net)
    with torch.no_grad():
        fg.coarse_geo_net.embed_fn.embeddings.copy_(torch.from_numpy(tbl))
        fg.obj_bound.volume.update_bitfield(torch.from_numpy(bf).to(gpu), ops='overwrite')
    return fg, w


@pytest.mark.parametrize('variant,net', [('k2', 'lin'), ('k2', 'nb'), ('tb', 'lin')])
def test_dense_module_path_reproduces_reference_run(gpu, g21, variant, net, monkeypatch):
    """build_model with the fixture's overrides (GeoNet / RadianceNet on the torch-semantics hash grid): FullModel.forward ->
    FgModel.forward -> dense (R, P') tensors like the reference, every op on the HIP kernels; reference state_dict loaded by name."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    import arcnerf_amd.geometry.volume as vol_mod
    g, tbl, bf = g21
    if variant == 'tb':   # the reference's torch AABB semantics (what its CPU / extension-less path computes)
        real = vol_mod.aabb_ray_intersection
        monkeypatch.setattr(vol_mod, 'aabb_ray_intersection', lambda o, d, r, eps=1e-7, force_torch=False: real(o, d, r, eps, True))
    ov = [str(v) for v in g['overrides_base']] + [str(v) for v in g['overrides_' + net]]
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ov)).to(gpu)
    pre = '{}_{}_'.format(variant, net)
    sd = {k[len(pre) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre + 'sd.')}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith(('embeddings', 'bitfield', 'opafield')) for k in missing), (missing, unexpected)
    fg, _ = _load_module_weights(m, g, variant, net, tbl, bf, gpu)
    assert not fg.packed_path_eligible()
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    sampler_rng(reset=True)
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask'}
    G.check_outputs(g, pre + 'infer_', out['rgb'][0].cpu().numpy(), out['depth'][0].cpu().numpy(), out['mask'][0].cpu().numpy(), train=False)
    fg.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    assert set(out.keys()) == {'rgb_coarse', 'depth_coarse', 'mask_coarse'}
    G.check_outputs(g, pre + 'train0_', out['rgb_coarse'][0].detach().cpu().numpy(), out['depth_coarse'][0].detach().cpu().numpy(),
                    out['mask_coarse'][0].detach().cpu().numpy())
    assert abs(float(m.get_dynamicbs_factor()) - float(g[pre + 'train0_dynamicbs_factor'])) < 1e-3 * float(g[pre + 'train0_dynamicbs_factor'])
    loss = ((out['rgb_coarse'] - inputs['img']) ** 2).mean() * 100.0
    loss.backward()
    for n_, p in m.named_parameters():
        if p.grad is None:
            assert (pre + 'train0_grad.' + n_) not in g.files, n_
        elif n_.endswith('embed_fn.embeddings'):
            G.check_table_grad(g, pre + 'train0_tgrad_', p.grad.cpu().numpy())
        else:
            ref = g[pre + 'train0_grad.' + n_]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, n_


def test_packed_module_path_reproduces_reference_run(gpu, g21):
    """configs/nerf_ngp.yaml UNCHANGED (FusedMLPGeoNet / FusedMLPRadianceNet, tcnn back-ends -> the packed kernels): loaded with the
    reference's bias-free nets through the zero-column map, it must render and differentiate like the reference's run."""
    from arcnerf_amd.models import build_model
    from arcnerf_amd.ops.volume_func import sampler_rng
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g, tbl, bf = g21
    variant, net = 'k2', 'nb_fused'
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ['--model.rays.add_inf_z', 'True'])).to(gpu)
    fg, w = _load_module_weights(m, g, variant, net, tbl, bf, gpu)
    assert fg.packed_path_eligible()
    with torch.no_grad():
        for net_mod, layers in ((fg.coarse_geo_net, w['geo']), (fg.coarse_radiance_net, w['rad'])):
            flat = np.concatenate([W.reshape(-1) for W, _ in layers])
            assert net_mod.layers.params.numel() == flat.shape[0]
            net_mod.layers.params.copy_(torch.from_numpy(flat))
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    pre = 'k2_nb_'
    sampler_rng(reset=True)
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    G.check_outputs(g, pre + 'infer_', out['rgb'][0].cpu().numpy(), out['depth'][0].cpu().numpy(), out['mask'][0].cpu().numpy(), train=False)
    fg.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    G.check_outputs(g, pre + 'train0_', out['rgb_coarse'][0].detach().cpu().numpy(), out['depth_coarse'][0].detach().cpu().numpy(),
                    out['mask_coarse'][0].detach().cpu().numpy())
    assert abs(float(m.get_dynamicbs_factor()) - float(g[pre + 'train0_dynamicbs_factor'])) < 1e-3 * float(g[pre + 'train0_dynamicbs_factor'])
    loss = ((out['rgb_coarse'] - inputs['img']) ** 2).mean() * 100.0
    loss.backward()
    G.check_table_grad(g, pre + 'train0_tgrad_', fg.coarse_geo_net.embed_fn.embeddings.grad.cpu().numpy())
    nets = {}
    for name, net_mod in (('geo', fg.coarse_geo_net), ('rad', fg.coarse_radiance_net)):
        dims, flat, o_ = net_mod.layers.dims, net_mod.layers.params.grad.cpu().numpy(), 0
        layers = []
        for i in range(len(dims) - 1):
            k = dims[i] * dims[i + 1]
            layers.append((flat[o_:o_ + k].reshape(dims[i + 1], dims[i]), None))
            o_ += k
        nets[name] = layers
    G.check_net_grads(g, variant, net, 'train0', nets)
