"""G18: NeuS on a hash grid (second-order path) from the reference, run in the build container only.

configs/neus_ngp_multivol.yaml of this repo (= the model block of the reference's capture_qqtiger_neusngp_multivol.yaml) with
what cannot run on CPU replaced: the occupancy-pruned volume (CUDA sampler) by the sphere bound of configs/models/neus.yaml, the
tcnn back-ends by the reference's torch back-ends, no background; the hash grid shrunk.  What it pins is the sdf net ON the hash
encoder with normals taken by autograd (create_graph) and the rgb + Eikonal loss reaching the table through them.  The edited
YAML travels inside the fixture so the test builds the very same model.
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.insert(1, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(2, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

from arcnerf.models import build_model  # noqa: E402
from common.utils.cfgs_utils import load_configs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


def edited_config():
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'configs', 'neus_ngp_multivol.yaml')))
    m = cfg['model']
    del m['background']
    m['obj_bound'] = {'sphere': {'radius': 1.5}}
    m['rays'].update({'n_sample': 32, 'n_importance': 32, 'radius_bound': 1.5})
    m['chunk_pts'] = 4096
    m['geometry']['encoder'].update({'backend': 'torch', 'side': 3.0, 'n_levels': 8, 'hashmap_size': 12, 'base_res': 4, 'max_res': 64})
    m['radiance']['encoder']['view'].update({'backend': 'torch'})
    return yaml.dump(cfg, default_flow_style=False)


def one_ulp_rerun(model, inputs, draws, res, keys=('rgb', 'depth', 'mask', 'normal')):
    """The training pass once more with every ray origin moved to the next float: -> (per-ray largest change of any output, gradients).
    NeuS on a ROUGH field (a random +-0.1 hash table, cells down to 1/64: |d sdf / d x| of several units) multiplies an ulp of the
    position by that slope and by the up-sampling sharpness 64..512 before it reaches the weights: how far the reference moves under
    that perturbation is how well any fp32 implementation can be expected to agree with it.  Rays that move by more than 2e-5 are
    not stored; the gradients' movement on the stored rays travels in the fixture (`ulperr.*`) and sets their bar."""
    from tie_probe import RandTape
    moved = dict(inputs)
    moved['rays_o'] = torch.nextafter(inputs['rays_o'], torch.full_like(inputs['rays_o'], 1e9))
    tape = RandTape(0)
    with tape.replay([t.clone() for t in draws]):
        r2 = model({k: v.clone() for k, v in moved.items()}, inference_only=False, cur_epoch=20000)
    eik = ((r2['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((r2['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    model.zero_grad()
    loss.backward()
    sens = None
    for k in keys:
        d = (r2[k].detach() - res[k].detach()).abs()
        d = d.reshape(d.shape[0] * d.shape[1], -1).amax(dim=1)
        sens = d if sens is None else torch.maximum(sens, d)
    return sens.numpy(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def main(pool=800, margin=2e-6, pos_noise=5e-6, ulp_tol=2e-5):
    """Inference pass on the deterministic lattice (outputs), training pass with `perturb: True` (the yaml's value) on taped uniforms;
    the stored rays are those of a pool whose inverse-CDF decisions all have a margin >= `margin` (tie_probe.py; why: see
    make_golden_fullwidth.neus)."""
    import arcnerf.models.neus_model as NM
    import arcnerf.render.ray_helper as RH
    from tie_probe import ProbeU, RandTape, inference_flip_sensitivity, inference_lattice_margin
    text = edited_config()
    with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
        f.write(text)
    torch.manual_seed(1818)
    model = build_model(load_configs(f.name, []), None)
    os.unlink(f.name)
    with torch.no_grad():   # the reference initialises the table in +-1e-4: give the encoder something to say
        emb = model.fg_model.geo_net.embed_fn.embeddings
        emb.copy_((torch.rand(emb.shape, generator=torch.Generator().manual_seed(1)) - 0.5) * 0.2)
    g = torch.Generator().manual_seed(1819)
    B, N = 2, 64
    o = torch.randn(1, pool, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 3.0
    d = -o + (torch.rand(1, pool, 3, generator=g) - 0.5) * 1.6
    d = d / d.norm(dim=-1, keepdim=True)
    n_miss = 8                                          # rays that pass the sphere bound by
    tang = torch.cross(o[0, -n_miss:], torch.tensor([0.0, 0.0, 1.0]).expand(n_miss, 3), dim=-1)
    d[0, -n_miss:] = tang / tang.norm(dim=-1, keepdim=True)
    pool_in = {'rays_o': o, 'rays_d': d, 'rays_r': torch.zeros(1, pool, 1), 'img': torch.rand(1, pool, 3, generator=g),
               'bkg_color': torch.rand(1, pool, 3, generator=g)}
    assert model.fg_model.get_ray_cfgs('perturb') is True
    hit = model.fg_model.obj_bound.get_near_far_from_rays({k: v[0] for k, v in pool_in.items()})[2].view(-1)
    n_hit = int(hit.sum())
    tape = RandTape(1820)
    model_res_holder = []
    with tape.record(), ProbeU(tape, RH, NM) as probe:
        model_res_holder.append(model({k: v.clone() for k, v in pool_in.items()}, inference_only=False, cur_epoch=20000))
    pool_draws = [t.clone() for t in tape.draws]
    m_pool = np.full(pool, 1.0)
    m_pool[hit.numpy()] = probe.per_ray()[1]
    assert all(t.shape[0] in (n_hit, pool) for t in pool_draws) and int((~hit[-n_miss:]).sum()) == n_miss
    pool_res = model_res_holder[0]
    ulp_sens, _ = one_ulp_rerun(model, pool_in, pool_draws, pool_res)
    print('rays whose outputs move by more than', ulp_tol, 'when the origin moves by one ulp:', int((ulp_sens >= ulp_tol).sum()), 'of', pool)
    flip = inference_flip_sensitivity(model, pool_in, RH)     # inference = deterministic lattice: u = 1.0 vs cdf[-1] is a coin flip
    print('rays whose inference outputs depend on the u = 1 decision:', int((flip >= 1e-5).sum()), 'of', pool)
    lat = np.full(pool, 1.0)
    noise = np.zeros(pool)
    lat[hit.numpy()], noise[hit.numpy()] = inference_lattice_margin(model, pool_in, RH, NM)
    m_pool = np.minimum(m_pool, lat)      # one margin for both passes: taped uniforms (training) and lattice points (inference)
    noise[hit.numpy()] = np.maximum(noise[hit.numpy()], probe.per_ray_noise())   # worst-conditioned sample of either pass (tie_probe.position_noise)
    ok = np.nonzero((m_pool >= margin) & hit.numpy() & (flip < 1e-5) & (noise <= pos_noise) & (ulp_sens < 0.5 * ulp_tol))[0]
    print('rays with a sample in an ill-conditioned bin (position noise >', pos_noise, '):', int((noise > pos_noise).sum()), 'of', pool)
    sel = np.sort(np.concatenate([ok[:B * N - n_miss], np.arange(pool - n_miss, pool)]))
    assert len(sel) == B * N, len(ok)
    rows = (np.cumsum(hit.numpy()) - 1)[sel][hit.numpy()[sel]]
    inputs = {k: v[0, sel].reshape(B, N, -1).contiguous() for k, v in pool_in.items()}
    draws = [t[sel] if t.shape[0] == pool else t[rows] for t in pool_draws]
    out = {'config_yaml': np.array(text), 'pool_size': np.array(pool), 'pool_sel': sel, 'tie_margin': m_pool[sel], 'flip_sensitivity': flip[sel], 'position_noise': noise[sel], 'ulp_sensitivity': ulp_sens[sel]}
    print('pool', pool, 'hit', n_hit, 'below margin', int((m_pool < margin).sum()), 'min margin kept', m_pool[sel].min())
    res = model({k: v.clone() for k, v in inputs.items()}, inference_only=True)
    for k, v in res.items():
        out['infer_' + k] = v.detach().numpy()
    with tape.replay(draws), ProbeU(tape, RH, NM) as probe:
        res = model({k: v.clone() for k, v in inputs.items()}, inference_only=False, cur_epoch=20000)
    assert probe.per_ray()[1].min() >= margin
    for i, t in enumerate(draws):
        out['draw_{:02d}'.format(i)] = t.numpy()
    eik = ((res['normal_pts'].norm(dim=-1) - 1.0) ** 2).mean()
    loss = ((res['rgb'] - inputs['img']) ** 2).mean() + 0.1 * eik
    model.zero_grad()
    loss.backward()
    out['train_loss'], out['train_eikonal'] = loss.detach().numpy(), eik.detach().numpy()
    for k, v in res.items():
        if torch.is_tensor(v):
            out['train_' + k] = v.detach().numpy()
    for k, v in inputs.items():
        out['in_' + k] = v.numpy()
    for k, v in model.state_dict().items():
        out['sd.' + k] = v.numpy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['grad.' + k] = p.grad.numpy()
    print('hit rays', int((res['mask'] > 0).sum()), 'of', B * N, 'loss', float(loss), 'eik', float(eik), 'table grad max',
          float(model.fg_model.geo_net.embed_fn.embeddings.grad.abs().max()), 'params with grad',
          [k for k, p in model.named_parameters() if p.grad is not None], 'draws', [tuple(t.shape) for t in draws])
    # the reference's own conditioning on the stored rays: the same pass with every ray origin moved by ONE ulp
    g32 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    sens, g_ulp = one_ulp_rerun(model, inputs, draws, res)
    assert sens.max() < ulp_tol, sens.max()
    for k, v in g_ulp.items():
        out['ulperr.' + k] = np.array(float((v - g32[k]).abs().max() / g32[k].abs().max()))
    print('one-ulp re-run: outputs move by at most', sens.max(), '; gradients (relative to max):',
          {k: round(float(out['ulperr.' + k]), 5) for k in g_ulp})
    path = os.path.join(OUT, 'g18_neus_ngp_model.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1024, 'KB')


if __name__ == '__main__':
    main()
