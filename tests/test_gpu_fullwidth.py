"""BASELINE configs 1 / 3 / 5 at their FULL widths (8x256 skip nets, 128/256-wide radiance nets) against runs of the reference's
FullModel: golden G22 (nerf.yaml), G23 (neus.yaml), G24 (hdrnerf.yaml), tests/golden/make_golden_fullwidth.py.  The configs are the
repo's copies of the reference yamls with NO width overrides; big matrices come from tests/seeded_weights.py, gradients of those are
compared through their stored summaries."""
import os

import numpy as np
import pytest
import torch

import seeded_weights as SW
from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
CFG = os.path.join(ROOT, 'configs')


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def close(a, b, rtol=2e-4, atol=2e-4):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _build(gpu, cfg_name, g):
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    m = build_model(load_configs(os.path.join(CFG, cfg_name + '.yaml'), [str(v) for v in g['overrides']])).to(gpu)
    sd = {k: torch.from_numpy(v) for k, v in SW.state_dict_from_fixture(g).items()}
    m.load_state_dict(sd)      # strict: same parameter / buffer names as the reference
    inputs = {k[3:]: torch.from_numpy(g[k]).to(gpu) for k in g.files if k.startswith('in_')}
    return m, inputs


def _check_grads(m, g, rtol, loose=(), loose_rtol=1e-2):
    checked = seeded = 0
    for n, p in m.named_parameters():
        r = loose_rtol if n.endswith(tuple(loose)) and loose else rtol
        if ('gsum.' + n + '.max') in g.files:
            SW.check_grad({k: g['gsum.' + n + '.' + k] for k in ('head', 'mod16', 'sum', 'abs', 'max', 'proj')}, p.grad.cpu().numpy(), rtol=r, name=n)
            seeded += 1
        elif ('grad.' + n) in g.files:
            ref = g['grad.' + n]
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= r * np.abs(ref).max() + 1e-7, n
            checked += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    return checked, seeded


def test_vanilla_nerf_full_width(gpu):
    """config 1: 63 -> 8 x 256 (skip at 4) -> 257, radiance 27+256 -> 128 -> 3, coarse + fine nets, 64 + 128 samples per ray."""
    g = load_golden('g22_nerf_fullwidth')
    m, inputs = _build(gpu, 'nerf', g)
    assert m.fg_model.coarse_geo_net.W == 256 and m.fg_model.coarse_radiance_net.W == 128
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask'}
    for k in out:
        close(out[k].cpu().numpy(), g['infer_' + k], 1e-4, 1e-4)
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    assert set(out.keys()) == {'rgb_coarse', 'depth_coarse', 'mask_coarse', 'rgb_fine', 'depth_fine', 'mask_fine'}
    for k in out:
        close(out[k].detach().cpu().numpy(), g['train_' + k], 1e-4, 1e-4)
    loss = ((out['rgb_fine'] - inputs['img']) ** 2).mean() + ((out['rgb_coarse'] - inputs['img']) ** 2).mean()
    assert abs(float(loss) - float(g['train_loss'])) < 1e-5
    loss.backward()
    checked, seeded = _check_grads(m, g, 1e-3)
    assert checked >= 20 and seeded >= 18


def test_neus_full_width(gpu):
    """config 3: sdf net 8 x 256 softplus(100), geometric init, weight norm, skip-reduce + 1/sqrt(2); radiance 4 x 256 'pvnf'; normals by
    autograd with create_graph (Eikonal term reaches the weights through the second derivative)."""
    g = load_golden('g23_neus_fullwidth')
    m, inputs = _build(gpu, 'neus', g)
    assert m.fg_model.geo_net.W == 256 and m.fg_model.radiance_net.W == 256
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'depth', 'mask', 'normal'}
    for k in ('rgb', 'depth', 'mask', 'normal'):
        close(out[k].detach().cpu().numpy(), g['infer_' + k])
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    for k in ('rgb', 'depth', 'mask', 'normal'):
        close(out[k].detach().cpu().numpy(), g['train_' + k])
    bad = np.abs(out['normal_pts'].detach().cpu().numpy() - g['train_normal_pts']) > 2e-4 + 2e-4 * np.abs(g['train_normal_pts'])
    assert bad.mean() < 1e-3, bad.mean()
    prm = out['params'][0] if isinstance(out['params'], list) else out['params']
    assert abs(prm['scale'] - float(g['train_scale'])) < 1e-3
    eik = ((out['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((out['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    assert abs(float(eik) - float(g['train_eikonal'])) < 1e-5 and abs(float(loss) - float(g['train_loss'])) < 1e-5
    loss.backward()
    # one of the 40 rays has an up-sampled position on the other side of a near-tie of the inverse CDF (tools/diag_neus_fullwidth.py:
    # every gradient agrees to <= 1.3e-4 of its max except the two matrices that multiply the 2^9-frequency position embedding, which
    # feel that single sample).  Which side the sample lands on depends on the last bits of the sdf values, i.e. on the arithmetic
    # variant: 0.3e-2 .. 2.4e-2 over {split, exact-f32} products x {libm, hardware exp2 / log2} softplus (all of them 1e-7 apart in sdf;
    # the reference's own CPU and GPU runs differ by as much).  Everything else is held at 1e-3.
    checked, seeded = _check_grads(m, g, 1e-3, loose=('geo_net.layers.0.weight_v', 'geo_net.layers.5.weight_v'), loose_rtol=3e-2)
    assert checked >= 20 and seeded >= 12 and 'grad.fg_model.inv_s' in g.files


def test_hdrnerf_full_width(gpu):
    """config 5: nerf.yaml nets + three 1 -> 128 -> 1 tone mappers on ln(exposure) + log radiance, LDR and HDR compositing passes."""
    g = load_golden('g24_hdrnerf_fullwidth')
    m, inputs = _build(gpu, 'hdrnerf', g)
    with torch.no_grad():
        out = m({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    assert set(out.keys()) == {'rgb', 'hdr', 'depth', 'mask'}
    for k in out:
        close(out[k].cpu().numpy(), g['infer_' + k])
    m.fg_model.set_ray_cfgs('perturb', False)
    m.fg_model.set_ray_cfgs('noise_std', 0.0)
    out = m({k: v.clone() for k, v in inputs.items()}, inference_only=False)
    keys = {a + '_' + b for a in ('rgb', 'hdr', 'depth', 'mask', 'unit_exp') for b in ('coarse', 'fine')}
    assert set(out.keys()) == keys
    for k in keys:
        close(out[k].detach().cpu().numpy(), g['train_' + k])
    unit = sum(((out['unit_exp_' + s] - 0.5) ** 2).mean() for s in ('coarse', 'fine'))
    loss = ((out['rgb_fine'] - inputs['img']) ** 2).mean() + ((out['rgb_coarse'] - inputs['img']) ** 2).mean() + 0.5 * unit
    assert abs(float(loss) - float(g['train_loss'])) < 1e-5
    loss.backward()
    checked, seeded = _check_grads(m, g, 1e-3)
    assert checked >= 30 and seeded >= 18
