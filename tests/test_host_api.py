"""CPU tests of the host-side mirror of the reference's plugin API: config loading + overrides, registries, builders,
chunk_processing, state_dict key compatibility with a reference-exported checkpoint, and that the product path refuses to
run without a GPU instead of silently falling back."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

from arcnerf_amd.models import build_model
from arcnerf_amd.models.base_modules import build_encoder, build_geo_model, build_radiance_model
from arcnerf_amd.utils.cfgs_utils import dict_to_obj, get_value_from_cfgs_field, load_configs, obj_to_dict, remap_value
from arcnerf_amd.utils.registry import BOUND_REGISTRY, ENCODER_REGISTRY, MODEL_REGISTRY, MODULE_REGISTRY, Registry
from arcnerf_amd.utils.torch_utils import chunk_processing

CFG = os.path.join(ROOT, 'configs')


def test_remap_value_typing():
    cases = {'None': None, 'true': True, 'False': False, '12': 12, '-3': -3, '0.5': 0.5, '-0.25': -0.25, '1e-1': 0.1, '1e3': 1000.0,
             '-2e-2': -0.02, "'abc'": 'abc', 'str(12)': '12', '[1, 2.5, a]': [1, 2.5, 'a'], '1,2': [1, 2], 'plain': 'plain'}
    for k, v in cases.items():
        assert remap_value(k) == v, k


def test_load_configs_with_cli_overrides():
    c = load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ['--model.rays.n_sample', '256', '--model.obj_bound.volume.n_grid', '32',
                                                          '--model.geometry.encoder.backend', 'torch', '--extra.flag', 'None'])
    assert c.model.rays.n_sample == 256 and c.model.obj_bound.volume.n_grid == 32
    assert c.model.geometry.encoder.backend == 'torch' and c.extra.flag is None
    assert c.model.geometry.out_act_cfg.type == 'TruncExp'
    assert get_value_from_cfgs_field(c.model.rays, 'near') is None
    assert obj_to_dict(dict_to_obj({'a': {'b': [1, 2]}})) == {'a': {'b': [1, 2]}}


def test_registries_hold_the_path_components():
    assert 'NeRF' in MODEL_REGISTRY
    for n in ('FreqEmbedder', 'HashGridEmbedder', 'SHEmbedder'):
        assert n in ENCODER_REGISTRY
    for n in ('GeoNet', 'RadianceNet', 'FusedMLPGeoNet', 'FusedMLPRadianceNet'):
        assert n in MODULE_REGISTRY
    for n in ('BasicBound', 'VolumeBound'):
        assert n in BOUND_REGISTRY
    r = Registry('t')

    @r.register()
    class A:
        pass

    assert r.get('A') is A
    with pytest.raises(KeyError):
        r.register(A)
    with pytest.raises(KeyError):
        r.get('B')


def test_builders_and_output_dims():
    enc, inp, nf = build_encoder(dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 10}))
    assert (enc.get_output_dim(), inp, nf) == (63, 3, 10)
    enc, _, _ = build_encoder(None)
    assert enc.get_output_dim() == 3
    enc, _, _ = build_encoder(dict_to_obj({'type': 'SHEmbedder', 'input_dim': 3, 'n_freqs': 4, 'include_input': False}))
    assert enc.get_output_dim() == 16
    enc, _, _ = build_encoder(dict_to_obj({'type': 'HashGridEmbedder', 'input_dim': 3, 'n_freqs': 0, 'side': 2.0, 'include_input': True,
                                           'n_levels': 4, 'hashmap_size': 8, 'base_res': 2, 'max_res': 16, 'unknown_key': 1}))
    assert enc.get_output_dim() == 4 * 2 + 3 and enc.embeddings.shape == (enc.n_total_embed, 2)
    c = load_configs(os.path.join(CFG, 'nerf_ngp.yaml'))
    geo, rad = build_geo_model(c.model.geometry), build_radiance_model(c.model.radiance)
    assert geo.layers.dims == [32, 64, 16] and rad.layers.dims == [32, 64, 64, 3] and rad.init_input_dim == 32
    assert [n for n, p in geo.named_parameters() if p.requires_grad] == ['embed_fn.embeddings', 'layers.params']
    assert geo.embed_fn.resolutions == [15, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    assert geo.embed_fn.n_total_embed == 6098108


def test_build_model_ngp_and_vanilla():
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml')))
    fg = m.get_fg_model()
    assert fg.packed_path_eligible() and fg.get_obj_bound_type() == 'volume'
    assert fg.get_render_cfgs('max_allowance') == 1 << 18 and fg.get_render_cfgs('depth_far') == 10.0
    sd = m.state_dict()
    for k in ('fg_model.obj_bound.volume.bitfield', 'fg_model.obj_bound.volume.opafield', 'fg_model.coarse_geo_net.embed_fn.embeddings'):
        assert k in sd
    assert sd['fg_model.obj_bound.volume.bitfield'].shape == (128, 128, 128) and bool(sd['fg_model.obj_bound.volume.bitfield'].all())
    m = build_model(load_configs(os.path.join(CFG, 'nerf.yaml')))
    assert not m.get_fg_model().packed_path_eligible()
    assert sum(p.numel() for p in m.parameters()) == 1191688  # two 8x256 + 1x128 stacks like the reference


def test_state_dict_keys_match_reference_checkpoint():
    g = load_golden('g9_nerf_model')
    ov = [str(v) for v in g['overrides']]
    m = build_model(load_configs(os.path.join(CFG, 'nerf.yaml'), ov))
    ref_keys = {k[3:]: g[k].shape for k in g.files if k.startswith('sd.')}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(ref_keys) == set(mine)
    for k, s in ref_keys.items():
        assert tuple(s) == mine[k], k


def test_chunk_processing_semantics():
    calls = []

    def f(a, d, s, n):
        calls.append(a.shape[0])
        return a * 2, {'x': d['x'] + 1, 'tag': s}, n, None

    a, x = torch.arange(10.0), np.arange(10.0)
    o = chunk_processing(f, 4, False, a, {'x': x, 'k': 'v'}, 'hello', 7)
    assert calls == [4, 4, 2]
    assert torch.equal(o[0], a * 2) and np.array_equal(o[1]['x'], x + 1)
    assert o[1]['tag'] == ['hello'] * 3 and o[2] == [7] * 3 and o[3] == [None] * 3
    assert torch.equal(chunk_processing(lambda t: t + 1, 0, False, a), a + 1)       # chunk_size <= 0: direct call
    assert chunk_processing(lambda s: s * 2, 4, False, 'ab') == 'abab'              # no array argument: direct call
    with pytest.raises(AssertionError):
        chunk_processing(lambda p, q: p, 4, False, torch.zeros(3), torch.zeros(4))
    # one chunk: its outputs ARE the result (no concatenation pass); several chunks: concatenated copies
    one = chunk_processing(lambda t: {'y': t + 1}, 16, False, a)
    assert torch.equal(one['y'], a + 1)
    keep = []
    res = chunk_processing(lambda t: keep.append(t + 1) or keep[-1], 16, False, a)
    assert res is keep[0]


def test_product_path_has_no_cpu_fallback():
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ['--model.obj_bound.volume.n_grid', '16', '--model.geometry.encoder.hashmap_size', '10']))
    inputs = {'rays_o': torch.zeros(1, 8, 3), 'rays_d': torch.ones(1, 8, 3), 'rays_r': torch.zeros(1, 8, 1), 'bkg_color': torch.zeros(1, 8, 3)}
    with pytest.raises(RuntimeError):
        m(inputs, inference_only=True)


def test_mask_compaction_and_fix_step_helpers_match_reference():
    """the reference's torch fallback helpers (a4'): pure tensor code, pinned on the reference's own outputs (G3)"""
    from arcnerf_amd.render.ray_helper import get_zvals_from_near_far_fix_step, handle_valid_mask_zvals
    g = load_golden('g3_zvals')
    z, m = handle_valid_mask_zvals(torch.from_numpy(g['hv_z']), torch.from_numpy(g['hv_m']))
    assert np.array_equal(m.numpy(), g['hv_m_out']) and np.array_equal(z.numpy(), g['hv_z_out'])
    zz, mm = get_zvals_from_near_far_fix_step(torch.tensor([[1.0], [2.0]]), torch.tensor([[1.35], [5.0]]), 0.1, 6)
    assert mm.tolist() == [[True, True, True, True, True, False], [True] * 6]
    np.testing.assert_allclose(zz[0].numpy(), [1.0, 1.1, 1.2, 1.3, 1.35, 1.35], rtol=1e-6)


def test_refresh_cell_selection_is_sync_free_and_matches_reference_sets():
    """uniform part: n/4 distinct cells; occupied part: the first n/4 occupied cells in flat order (volume_bound.py:178-190)"""
    from arcnerf_amd.geometry.volume import select_refresh_cells
    n = 16 ** 3
    g = torch.Generator().manual_seed(0)
    for frac in (0.05, 0.6):
        bf = torch.rand(n, generator=g) < frac
        cells, n_valid = select_refresh_cells(bf, n, {}, np.random.default_rng(1))
        n_s = n // 4
        occ = torch.nonzero(bf)[:, 0]
        want = occ[:n_s]
        assert int(n_valid) == n_s + want.numel()
        uni = cells[:n_s]
        assert uni.unique().numel() == n_s and int(uni.min()) >= 0 and int(uni.max()) < n
        assert torch.equal(cells[n_s:int(n_valid)], want)


def test_sphere_bound_is_registered_and_built_from_cfgs():
    """build_obj_bound picks SphereBound for an `obj_bound.sphere` block (obj_bound/__init__.py:25-62) with the reference's
    Sphere accessors; the ray test itself needs the GPU (tests/test_gpu_kernels.py)."""
    from arcnerf_amd.models.base_modules.obj_bound import build_obj_bound
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    from arcnerf_amd.utils.registry import BOUND_REGISTRY
    assert 'SphereBound' in BOUND_REGISTRY
    bound, kind = build_obj_bound(dict_to_obj({'obj_bound': {'sphere': {'origin': [0.0, 0.5, 0.0], 'radius': 2.0}, 'epoch_optim': 16}}))
    assert kind == 'sphere' and type(bound).__name__ == 'SphereBound'
    assert bound.get_obj_bound().get_radius(in_float=True) == 2.0
    assert bound.get_obj_bound().get_origin(in_tuple=True) == (0.0, 0.5, 0.0)
    assert bound.get_optim_cfgs('epoch_optim') == 16
    with pytest.raises(RuntimeError):   # no CPU fallback on the product path
        bound.get_near_far_from_rays({'rays_o': torch.zeros(4, 3), 'rays_d': torch.ones(4, 3)})


def test_neus_yaml_builds_and_loads_reference_state_dict():
    """configs/models/neus.yaml (the reference's file) builds the Neus mirror with SphereBound, geometric init, weight norm and
    softplus(100); a state_dict exported by the reference loads with strict=True (same parameter names and shapes)."""
    import numpy as np
    from conftest import ROOT, load_golden
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    g = load_golden('g13_neus_model')
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'neus.yaml'), [str(v) for v in g['overrides']]))
    fg = m.fg_model
    assert type(fg).__name__ == 'Neus' and type(fg.obj_bound).__name__ == 'SphereBound' and fg.sigma_reverse()
    assert fg.get_ray_cfgs('n_importance') == 32 and fg.get_ray_cfgs('n_iter') == 4 and fg.radius_bound == 1.5
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}
    m.load_state_dict(sd, strict=True)
    assert abs(float(fg.forward_scale()) - float(g['train_scale'])) < 1e-4
    assert fg.get_cos_anneal(25000) == 0.5 and fg.get_cos_anneal(10 ** 6) == 1.0


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    """model_io: the `*.pt.tar` dict of common/utils/model_io.py - save, partial load with a size mismatch, resume of epoch/optimizer"""
    import torch
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj, load_configs
    from arcnerf_amd.utils.model_io import load_model, save_model
    ov = ['--model.geometry.W', '32', '--model.geometry.W_feat', '32', '--model.radiance.W', '16', '--model.radiance.W_feat_in', '32']
    m = build_model(load_configs(os.path.join(CFG, 'nerf.yaml'), ov))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    run = dict_to_obj({'dist': {'world_size': 1}, 'progress': {'start_epoch': -1}})
    path = save_model(None, m, opt, 7, 0.5, str(tmp_path), run)
    assert path.endswith('model_epoch007.pt.tar')
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'epoch', 'state_dict', 'optimizer', 'loss'} and not any(k.startswith('module.') for k in ck['state_dict'])
    m2 = build_model(load_configs(os.path.join(CFG, 'nerf.yaml'), ov))
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    load_model(None, m2, opt2, path, run, strict=True)
    assert run.progress.start_epoch == 7
    for (k, a), (_, b_) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b_), k
    # a model with another width takes what fits and reports the rest
    m3 = build_model(load_configs(os.path.join(CFG, 'nerf.yaml'), ov[:-1] + ['32', '--model.radiance.W', '24']))
    log = []
    logger = type('L', (), {'add_log': lambda self, msg, level='info': log.append((level, msg))})()
    load_model(logger, m3, None, save_model(None, m, opt, 1, 0.0, str(tmp_path), run, spec_name='final'), dict_to_obj({'dist': {'world_size': 1}}))
    assert any(lv == 'warning' and 'size mismatch' in msg for lv, msg in log)
    assert torch.equal(m3.state_dict()['fg_model.coarse_geo_net.layers.0.weight'], m.state_dict()['fg_model.coarse_geo_net.layers.0.weight'])


def test_refresh_cell_subset_is_a_bijection_and_spatially_uniform():
    """select_refresh_cells draws its n/4 'uniform' cells as the image of range(n/4) under a seeded bijection of the cell range
    (the reference uses torch.randperm): every cell exactly once over the full range, and the quarter subset spreads over the grid
    like a uniform draw (octants and x / y / z slabs within 5 sigma of the binomial expectation) for many seeds."""
    import numpy as np
    import torch
    from arcnerf_amd.geometry.volume import mix_permutation
    n_grid = 32
    n = n_grid ** 3
    ar = torch.arange(n)
    for seed in range(20):
        rng = np.random.default_rng(seed)
        full = mix_permutation(ar, n, rng).numpy()
        assert np.array_equal(np.sort(full), np.arange(n))
        rng = np.random.default_rng(seed)
        sub = mix_permutation(ar[:n // 4], n, rng).numpy()
        assert np.array_equal(sub, full[:n // 4])
        x, y, z = sub // (n_grid * n_grid), (sub // n_grid) % n_grid, sub % n_grid
        octant = (x >= n_grid // 2) * 4 + (y >= n_grid // 2) * 2 + (z >= n_grid // 2)
        cnt = np.bincount(octant, minlength=8)
        exp, sd = n / 32, np.sqrt(n / 4 * (1 / 8) * (7 / 8))
        assert np.abs(cnt - exp).max() < 5 * sd, (seed, cnt)
        for axis in (x, y, z):
            c = np.bincount(axis, minlength=n_grid)
            e, s_ = n / 4 / n_grid, np.sqrt(n / 4 / n_grid)
            assert np.abs(c - e).max() < 5 * s_, (seed, c)


def test_morton_order_of_the_synthetic_cascade_matches_the_oracle(oracle):
    """pipeline.morton3d (used to lay synthetic occupancy into a MultiVol cascade for bench.py) is the Morton code of the oracle / kernels"""
    import numpy as np
    from arcnerf_amd.pipeline import morton3d, synthetic_cascade_bits
    rng = np.random.default_rng(0)
    xyz = rng.integers(0, 128, size=(4096, 3)).astype(np.uint32)
    assert np.array_equal(morton3d(xyz[:, 0], xyz[:, 1], xyz[:, 2]).astype(np.int64), oracle.morton3d(xyz).astype(np.int64))
    bits = synthetic_cascade_bits(16, 2, 0.1, seed=0)
    assert bits.shape == (2 * 16 ** 3 // 8,) and 0.08 < np.unpackbits(bits).mean() < 0.3


def test_dynamic_batch_size_ring_matches_the_reference_arithmetic():
    """FgModel.adjust_dynamicbs_factor keeps each step's valid-sample count in a device ring (one copy per step, no host read) and
    get_dynamicbs_factor does the reference's arithmetic on read-back (arcnerf/models/fg_model.py:105-128: a Python float accumulating
    max_allowance / (n + 1), divided by the number of measurements, then reset) - also when the ring wraps before anybody asks."""
    m = build_model(load_configs(os.path.join(CFG, 'nerf_ngp.yaml'), ['--model.obj_bound.volume.n_grid', '16']))
    fg = m.fg_model
    cap = float(fg.render_cfgs['max_allowance'])
    assert cap > 0
    g = torch.Generator().manual_seed(0)
    counts = torch.randint(1, 400000, (37,), generator=g)
    for c in counts:
        fg.adjust_dynamicbs_factor(n_valid=c.to(torch.int32).reshape(1)[0])
    ref = sum(cap / (float(c) + 1.0) for c in counts.tolist()) / len(counts)
    assert fg.get_dynamicbs_factor() == ref
    assert fg.render_cfgs['measured_count'] == 0 and fg.get_dynamicbs_factor() == 1       # reset; nothing measured -> 1
    mask = torch.zeros(5, 7, dtype=torch.bool)
    mask[1, :3] = True
    fg.adjust_dynamicbs_factor(mask_pts=mask)                                               # the reference's signature: a sample mask
    assert fg.get_dynamicbs_factor() == cap / 4.0
    fg._DYNBS_RING = 8                                                                      # wrap: drained into the running sum
    fg._dynbs_ring = None
    for c in counts:
        fg.adjust_dynamicbs_factor(n_valid=c.reshape(1)[0])
    got = fg.get_dynamicbs_factor()
    assert abs(got - ref) <= 1e-12 * ref
    fg.render_cfgs['max_allowance'] = -1                                                    # switched off: nothing is recorded
    fg.adjust_dynamicbs_factor(n_valid=counts[0])
    assert fg.render_cfgs['measured_count'] == 0


def test_host_value_caches_follow_their_tensors():
    """Volume.get_diag_len / get_len / get_voxel_size and Sphere radius / origin hand out Python numbers read back once per VERSION of
    the tensor they come from (each read of a device scalar is a host / device synchronisation; the samplers ask every step): in-place
    writers, re-registered buffers and the setters must all be seen; scalar_tensor keeps one device tensor per value."""
    from arcnerf_amd.geometry.sphere import Sphere
    from arcnerf_amd.geometry.volume import Volume
    from arcnerf_amd.ops import functional as F
    v = Volume(n_grid=8, origin=(0.0, 0.0, 0.0), side=2.0)
    assert abs(v.get_diag_len() - 12.0 ** 0.5) < 1e-6 and v.get_len() == (2.0, 2.0, 2.0) and v.get_voxel_size() == (0.25, 0.25, 0.25)
    first = v._host_cache['diag']
    assert v.get_diag_len() == first[1] and v._host_cache['diag'] is first                  # served from the cache
    with torch.no_grad():
        v.range.mul_(2.0)                                                                   # an in-place writer moves the version
    assert abs(v.get_diag_len() - 2.0 * 12.0 ** 0.5) < 1e-6 and v.get_voxel_size() == (0.5, 0.5, 0.5)
    with torch.no_grad():
        v.xyz_len.fill_(4.0)
    v.cal_range()                                                                           # a re-registered buffer drops the cache
    assert abs(v.get_diag_len() - 48.0 ** 0.5) < 1e-6 and v.get_len() == (4.0, 4.0, 4.0)
    v.set_n_grid(16)
    assert v.get_voxel_size() == (0.25, 0.25, 0.25)
    s = Sphere(origin=(0.0, 0.0, 0.0), radius=1.5)
    assert s.get_radius(in_float=True) == 1.5 and s.get_origin(in_tuple=True) == (0.0, 0.0, 0.0)
    s.set_radius(2.5)
    s.set_origin((1.0, 2.0, 3.0))
    assert s.get_radius(in_float=True) == 2.5 and s.get_origin(in_tuple=True) == (1.0, 2.0, 3.0)
    a, b = F.scalar_tensor(64.0, 'cpu'), F.scalar_tensor(64, 'cpu')
    assert a is b and float(a) == 64.0 and F.scalar_tensor(128.0, 'cpu') is not a


def test_trainer_mirror_rules_on_cpu():
    """arcnerf_amd.trainer: the dynamic batch size (trainer/pipeline.py:222-241: only when epoch % update_epoch == 0 AND epoch > 500; float
    div_round_up to a multiple of 128; capped), the measurement behind it (fg_model.py:105-130), the reference's Huber (loss/img_loss.py:80-100:
    torch's divided by delta), AllLoss' dict, and EMA.ema_step's de-biased write-back (trainer/ema.py:29-43) with set_n_step."""
    import torch
    from arcnerf_amd import trainer as T
    m = T.DynamicBsMeter(1 << 15)
    for n in (52005, 47096, 51134):
        m.add(n)
    want = sum(float(1 << 15) / (float(n) + 1) for n in (52005, 47096, 51134)) / 3
    p = T.Pipeline()
    p.set_info('n_rays', 256)
    p.set_info('dynamic_batch_size', 4)
    p.set_info('dynamic_max_batch_size', 1024)
    assert p.fetch_step_update_dynamic_bs(500, m) == 256 and m.measured_count == 3       # epoch > 500 is strict: nothing read, nothing reset
    assert p.fetch_step_update_dynamic_bs(503, m) == 256
    n = p.fetch_step_update_dynamic_bs(504, m)
    assert n == min(int((256 * want + 127) // 128 * 128), 1024) == 256 and m.measured_count == 0
    m.add(3000)
    assert p.fetch_step_update_dynamic_bs(508, m) == 1024                                  # 256 * 10.9 -> capped
    assert T.DynamicBsMeter(-1).factor() == 1
    # Huber / ImgLoss / AllLoss
    x, y = torch.tensor([[0.0, 0.05, 0.5]]), torch.tensor([[0.0, 0.0, 0.0]])
    h = T.HuberLoss(0.1)(x, y)
    assert torch.allclose(h, torch.tensor([[0.0, 0.5 / 0.1 * 0.05 ** 2, 0.5 - 0.05]]))
    assert torch.allclose(h, torch.nn.functional.huber_loss(x, y, delta=0.1, reduction='none') / 0.1)
    cfg = type('C', (), {})()
    cfg.loss = type('C', (), {})()
    cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    out = T.build_loss(cfg)({'img': y[None]}, {'rgb_coarse': x[None]})
    assert out['names'] == ['ImgLoss'] and abs(float(out['sum']) - 3000.0 * float(h.mean())) < 1e-3
    # EMA (the eager form, any optimiser): three steps from n_step 496
    lin = torch.nn.Linear(3, 2)
    ema = T.EMA(lin, 0.95)
    ema.set_n_step(496)
    old = {k: v.clone() for k, v in lin.named_parameters()}
    for step in range(3):
        with torch.no_grad():
            for q in lin.parameters():
                q.add_(0.1 * (step + 1))
        cur = {k: v.clone() for k, v in lin.named_parameters()}
        ema.ema_step()
        n_ = 497 + step
        for k, q in lin.named_parameters():
            ref = ((1 - 0.95) * cur[k] + 0.95 * old[k] * (1 - 0.95 ** (n_ - 1))) * (1.0 / (1 - 0.95 ** n_))
            assert torch.allclose(q, ref, atol=1e-6)
            old[k] = ref


def test_fused_step_refuses_what_it_does_not_implement_and_train_epoch_draws_ahead_only_when_it_commutes():
    """trainer.FusedNgpStep.why_not names the reason for every combination that is not the NGP recipe (no silent detour); trainer.train_epoch
    hands a stepper the batch of epoch + 1 only when neither the bound's refresh nor the batch-size rule acts at that epoch"""
    import os
    import torch
    from arcnerf_amd import trainer as T
    from arcnerf_amd.models import build_model
    from arcnerf_amd.utils.cfgs_utils import load_configs
    from conftest import ROOT
    m = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf_ngp.yaml'), ['--model.obj_bound.volume.n_grid', '16']))
    cfg = type('C', (), {})()
    cfg.loss = type('C', (), {})()
    cfg.loss.ImgLoss = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    lf = T.build_loss(cfg)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-2)
    assert 'FusedAdam' in T.FusedNgpStep.why_not(m, lf, opt)
    with pytest.raises(RuntimeError, match='FusedAdam'):
        T.FusedNgpStep(m, lf, opt)
    m2 = build_model(load_configs(os.path.join(ROOT, 'configs', 'nerf.yaml'), []))
    assert 'packed instant-ngp' in T.FusedNgpStep.why_not(m2, lf, opt)

    # train_epoch's look-ahead: a stand-in stepper (two batches in flight) records what it is handed
    class Stepper:
        def __init__(self):
            self.calls, self._queue = [], []

        def can_run_ahead(self, epoch):
            return epoch % 4 != 0            # (the bound refreshes every 4 epochs)

        def ahead_room(self):
            return 2 - len(self._queue)

        def next_ahead_epoch(self, epoch):
            return self._queue[-1][0] + 1 if self._queue else epoch + 1

        def hold_ahead(self, epoch, feed_in):
            self._queue.append((epoch, feed_in))

        def take_ahead(self, epoch):
            if self._queue and self._queue[0][0] == epoch:
                return self._queue.pop(0)[1]
            self._queue = []
            return None

        def ahead(self):
            return [f for _, f in self._queue]

        def __call__(self, feed_in, epoch, next_feed_in=None):
            self.calls.append((epoch, feed_in, list(next_feed_in or [])))
            return {}, {'sum': 0.0}

    class Model:
        def __init__(self):
            self.optimized = []

        def optimize(self, epoch):
            self.optimized.append(epoch)

        def get_dynamicbs_factor(self):
            return 2.0

    tp = T.Pipeline()
    tp.set_info('n_rays', 128)
    tp.set_info('dynamic_batch_size', 4)
    tp.set_info('dynamic_max_batch_size', 1024)
    st, mdl, drawn = Stepper(), Model(), []

    def get_batch(n):
        drawn.append(n)
        return ('batch', len(drawn), n)

    for epoch in range(500, 509):
        T.train_epoch(mdl, get_batch, lf, None, None, tp, epoch, total_epoch=509, stepper=st)
    # every epoch's batch is drawn exactly once, in order, with the ray count the rule gives AT that epoch (x 2 at 504 and 508: epoch > 500)
    assert [c[1][1] for c in st.calls] == list(range(1, 10))
    assert drawn == [128, 128, 128, 128, 256, 256, 256, 256, 512]
    # handed ahead: up to two, never across an epoch at which the refresh or the rule acts, never past the end of the run
    assert [[b[1] for b in c[2]] for c in st.calls] == [[2, 3], [3, 4], [4], [], [6, 7], [7, 8], [8], [], []]
    # model.optimize ran for every epoch whose batch was not drawn early (where it is a no-op by the stepper's word)
    assert mdl.optimized == [500, 504, 508]


def test_fused_adam_state_dict_refuses_stale_sharded_moments():
    """Under distributed.ShardedGradSync a rank's Adam moments are current for its own shard only: state_dict() must not hand out a state
    whose other shards are stale - it asks for the collective gather first (FusedAdam.gather_sharded_state on every rank)."""
    import pytest
    import torch
    from arcnerf_amd import distributed as D
    from arcnerf_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(8))
    opt = FusedAdam([p], lr=1e-3)
    sync = D.ShardedGradSync(8, world=2, rank=0)
    opt.shard_sync = sync
    assert 'state' in opt.state_dict()              # nothing stepped yet: current
    sync.launch(torch.zeros(8))                      # (no process group: no collective is issued; the step marks the moments as sharded)
    assert not sync.moments_current
    with pytest.raises(RuntimeError, match='gather_sharded_state'):
        opt.state_dict()
    sync.gather_moments(torch.zeros(8), torch.zeros(8))
    assert sync.moments_current and 'state' in opt.state_dict()


def test_no_name_in_the_package_is_read_without_being_bound_anywhere():
    """A removed module-level switch must not leave a reader behind (a NameError that only the GPU suite would meet):
    tools/check_undefined_names.py walks every module of the package, the bench and the entry points."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_undefined_names.py'), os.path.join(root, 'arcnerf_amd'),
                          os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py'), os.path.join(root, 'oracle')],
                         capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == '', out.stdout + out.stderr
