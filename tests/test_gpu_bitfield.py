"""GPU parity of the `_bitfield_func` family (K5-K10) and of BitfieldBound, through the C ABI, against the CPU oracle.
Everything here is index / byte / bit-pattern work or a sampler whose t values repeat the oracle's IEEE op sequence: the
bar is bit-exact.  (No reference-produced vectors exist for these CUDA-only kernels: see tests/test_oracle_bitfield.py.)"""
import numpy as np
import pytest
import torch

from test_gpu_kernels import _blob_bitfield, _rays, dev, host
from test_oracle_bitfield import to_morton_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from arcnerf_amd.ops import functional
    return functional


@pytest.mark.parametrize('n_grid,n_pts,R', [(16, 256, 500), (128, 1024, 4096)])
def test_k5_sampler_bit_exact_vs_oracle(F, oracle, n_grid, n_pts, R):
    rng = np.random.default_rng(n_grid + 1)
    bf = _blob_bitfield(n_grid, rng, 0.15 if n_grid == 16 else 0.05)
    bits = to_morton_bits(bf, oracle)
    o, d = _rays(rng, R)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, _ = oracle.aabb_intersection(o, d, aabb23[None])
    dt = np.float32(2 * np.sqrt(3.0) / n_pts)
    h = oracle.Pcg32(9121)
    h.advance()
    z_ref, m_ref, c_ref = oracle.sparse_volume_sampling_bit(o, d, near, far, n_pts, dt, aabb23, n_grid, bits, 0.2, h.state, h.inc)
    z, m, c = F.sparse_volume_sampling_bit(dev(o), dev(d), dev(near), dev(far), n_pts, float(dt), dev(aabb23), n_grid, dev(bits),
                                           0.2, h.state, h.inc, want_counts=True)
    assert np.array_equal(host(c), c_ref) and np.array_equal(host(m), m_ref)
    assert np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32))
    assert c_ref.sum() > 10 * R / 4
    # the wave-per-ray compacted marcher over the Morton layout (bitfield mode 2) == compaction of the dense form
    pk = F.march_packed(dev(o), dev(d), dev(aabb23), n_grid, dev(bits), n_pts, float(dt), 0.2, h.state, h.inc, packed_bits=2)
    total = int(pk['offsets'][-1].item())
    assert total == int(c_ref.sum()) and np.array_equal(host(pk['counts']), c_ref)
    assert np.array_equal(host(pk['t'][:total]).view(np.uint32), z_ref[m_ref].view(np.uint32))


def test_k5_edge_cases(F, oracle):
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    n_grid, n_pts = 4, 8
    full = np.full(n_grid ** 3 // 8, 255, np.uint8)
    h = oracle.Pcg32(9121)
    # on the +x face (clamped voxel coordinate), a ray that misses, far < near, n_pts exhausted
    o = np.array([[1.0, -0.9, 0.1], [3.0, 3.0, 3.0], [0.0, 0.0, 0.0], [-0.99, 0.0, 0.0]], np.float32)
    d = np.array([[0.0, 1.0, 0.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [1.0, 0.0, 0.0]], np.float32)
    near = np.array([[0.0], [0.0], [0.9], [0.0]], np.float32)
    far = np.array([[1.5], [0.0], [0.1], [1.9]], np.float32)
    for bits in (full, np.zeros_like(full)):
        ref = oracle.sparse_volume_sampling_bit(o, d, near, far, n_pts, 0.1, aabb23, n_grid, bits, 0.0, h.state, h.inc)
        got = F.sparse_volume_sampling_bit(dev(o), dev(d), dev(near), dev(far), n_pts, 0.1, dev(aabb23), n_grid, dev(bits), 0.0,
                                           h.state, h.inc, want_counts=True)
        assert np.array_equal(host(got[2]), ref[2]) and np.array_equal(host(got[1]), ref[1])
        assert np.array_equal(host(got[0]).view(np.uint32), ref[0].view(np.uint32))
    assert ref[2].sum() == 0   # empty bitfield: nothing sampled
    # zero rays
    z, m = F.sparse_volume_sampling_bit(dev(o[:0]), dev(d[:0]), dev(near[:0]), dev(far[:0]), n_pts, 0.1, dev(aabb23), n_grid,
                                        dev(full), 0.0, h.state, h.inc)
    assert z.shape == (0, n_pts) and m.shape == (0, n_pts)
    with pytest.raises(RuntimeError):   # n_grid must be a power of two (Morton)
        F.sparse_volume_sampling_bit(dev(o), dev(d), dev(near), dev(far), n_pts, 0.1, dev(aabb23), 6, dev(np.zeros(27, np.uint8)),
                                     0.0, h.state, h.inc)


@pytest.mark.parametrize('n_grid,n', [(16, 3000), (128, 128 ** 3 // 4)])
def test_k6_grid_samples_bit_exact(F, oracle, n_grid, n):
    rng = np.random.default_rng(n_grid)
    grid = (rng.random(n_grid ** 3).astype(np.float32) - 0.5)
    h = oracle.Pcg32(9121)
    for step, thresh in ((0, -0.01), (5, 0.3), (70000, 0.49)):
        pos_ref, idx_ref = oracle.generate_grid_samples(grid, n, step, n_grid, thresh, h.state, h.inc)
        pos, idx = F.generate_grid_samples(dev(grid), n, step, n_grid, thresh, h.state, h.inc)
        assert np.array_equal(host(idx), idx_ref)
        assert np.array_equal(host(pos).view(np.uint32), pos_ref.view(np.uint32))
        h.advance()
    pos, idx = F.generate_grid_samples(dev(grid), 0, 0, n_grid, 0.0, h.state, h.inc)   # empty request: no launch
    assert pos.shape == (0, 3) and idx.shape == (0,)


def test_k7_to_k10_bit_exact(F, oracle):
    n_grid = 64
    n = n_grid ** 3
    rng = np.random.default_rng(7)
    idx = rng.integers(0, n, size=n // 2).astype(np.int32)
    idx[:1000] = 77   # heavy collisions on one cell
    den = rng.random(n // 2).astype(np.float32)
    tmp_ref = np.zeros(n, np.float32)
    oracle.splat_grid_samples(den, idx, tmp_ref)
    tmp = torch.zeros(n, device='cuda')
    F.splat_grid_samples(dev(den), dev(idx), n // 2, tmp)
    assert np.array_equal(host(tmp), tmp_ref)
    grid_ref = (rng.random(n).astype(np.float32) - 0.3).astype(np.float32)
    grid = dev(grid_ref.copy())
    oracle.ema_grid_samples_nerf(tmp_ref, grid_ref, 0.95)
    F.ema_grid_samples_nerf(tmp, grid, n, 0.95)
    assert np.array_equal(host(grid).view(np.uint32), grid_ref.view(np.uint32))
    # ragged element count (not a multiple of 4): only the first n_el cells move
    g2_ref = np.ones(11, np.float32)
    g2 = dev(g2_ref.copy())
    oracle.ema_grid_samples_nerf(np.zeros(11, np.float32), g2_ref[:7], 0.5)
    F.ema_grid_samples_nerf(torch.zeros(11, device='cuda'), g2, 7, 0.5)
    assert np.array_equal(host(g2), g2_ref)
    mean = float(np.clip(grid_ref, 0, None).mean())
    for thres in (0.01, 10.0):
        bf_ref = oracle.update_bitfield(grid_ref, mean, thres, n_grid)
        bf = torch.zeros(n // 8, dtype=torch.uint8, device='cuda')
        F.update_bitfield(grid, mean, bf, thres, n_grid)
        assert np.array_equal(host(bf), bf_ref)
        bf2 = torch.zeros(n // 8, dtype=torch.uint8, device='cuda')
        F.update_bitfield(grid, torch.tensor([mean], device='cuda'), bf2, thres, n_grid)   # device-side mean
        assert np.array_equal(host(bf2), bf_ref)
        assert float(F.count_bitfield(bf, n_grid)[0].item()) == oracle.count_bitfield(bf_ref, n_grid)
    sparse = np.zeros(n // 8, np.uint8)
    sparse[[1, 5, 9, n // 8 - 1]] = [1, 255, 16, 128]
    assert float(F.count_bitfield(dev(sparse), n_grid)[0].item()) == 32.0 == oracle.count_bitfield(sparse, n_grid)


def test_compat_bitfield_func_module(oracle):
    """`import _bitfield_func` (pybind signatures) through the compat path + the module-level generator bookkeeping"""
    import arcnerf_amd.compat as compat
    compat.install()
    import _bitfield_func as B
    from arcnerf_amd.ops.bitfield_func import bitfield_rng
    n_grid, n = 16, 1000
    rng = np.random.default_rng(3)
    grid = (rng.random(n_grid ** 3).astype(np.float32) - 0.5)
    bitfield_rng(reset=True)
    h = oracle.Pcg32(9121)
    for call in range(2):   # the file-static generator jumps 2^32 after every launch
        pos = torch.empty((n, 3), device='cuda')
        idx = torch.empty((n,), dtype=torch.int32, device='cuda')
        B.generate_grid_samples(dev(grid), 3, n, n_grid, 0.1, pos, idx)
        pos_ref, idx_ref = oracle.generate_grid_samples(grid, n, 3, n_grid, 0.1, h.state, h.inc)
        h.advance()
        assert np.array_equal(host(idx), idx_ref) and np.array_equal(host(pos).view(np.uint32), pos_ref.view(np.uint32))
    bf = _blob_bitfield(n_grid, rng, 0.15)
    bits = to_morton_bits(bf, oracle)
    o, d = _rays(rng, 64)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, _ = oracle.aabb_intersection(o, d, aabb23[None])
    z = torch.zeros((64, 32), device='cuda')
    m = torch.zeros((64, 32), dtype=torch.bool, device='cuda')
    B.sparse_volume_sampling_bit(dev(o), dev(d), dev(near), dev(far), 32, 0.1, dev(aabb23), n_grid, dev(bits), 0.0, z, m)
    z_ref, m_ref, _ = oracle.sparse_volume_sampling_bit(o, d, near, far, 32, 0.1, aabb23, n_grid, bits, 0.0, h.state, h.inc)
    assert np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32)) and np.array_equal(host(m), m_ref)
    with pytest.raises(RuntimeError):
        B.sparse_volume_sampling_bit(dev(o), dev(d), dev(near), dev(far), 32, 0.1, dev(aabb23), n_grid, dev(bits[:-1]), 0.0, z, m)
    with pytest.raises(RuntimeError):
        B.splat_grid_samples(dev(grid), torch.zeros(4, dtype=torch.int64, device='cuda'), 4, dev(grid))
    counter = torch.zeros(1, device='cuda')
    B.count_bitfield(dev(bits), counter, n_grid)
    assert float(counter.item()) == oracle.count_bitfield(bits, n_grid)
    bitfield_rng(reset=True)


def test_bitfield_bound_refresh_matches_oracle_flow(oracle):
    """BitfieldBound.optimize (warm-up: every cell; afterwards n/4 + n/4) == the same steps driven through the oracle"""
    from arcnerf_amd.models.base_modules.obj_bound import build_obj_bound
    from arcnerf_amd.ops.bitfield_func import bitfield_rng
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    n_grid = 16
    cfgs = dict_to_obj({'obj_bound': {'bitfield': {'n_grid': n_grid, 'side': 2.0}, 'epoch_optim': 2, 'epoch_optim_warmup': 4,
                                      'near_distance': 0.1, 'opa_thres': 0.01}})
    bound, kind = build_obj_bound(cfgs)
    assert kind == 'bitfield'
    bound = bound.cuda()
    bitfield_rng(reset=True)
    h = oracle.Pcg32(9121)

    def opacity_np(dt, pts):   # a blob of density around the origin
        r2 = (pts.astype(np.float64) ** 2).sum(-1)
        return (1.0 - np.exp(-np.exp(4.0 - 30.0 * r2) * dt)).astype(np.float32)

    seen = {}

    def opacity_gpu(dt, pts):
        seen['pts'] = pts.detach().cpu().numpy()
        return dev(opacity_np(dt, seen['pts']))

    n_el = n_grid ** 3
    grid = np.zeros(n_el, np.float32)
    step = 0
    bound.optimize(1, 64, opacity_gpu)   # not a refresh epoch
    assert step == bound.ema_step == 0 and int(bound.density_bitfield.min()) == 255
    for epoch in (2, 4, 6):
        bound.optimize(epoch, 64, opacity_gpu)
        n_u, n_n = (n_el, 0) if epoch < 4 else (n_el // 4, n_el // 4)
        pos_u, idx_u = oracle.generate_grid_samples(grid, n_u, step, n_grid, -0.01, h.state, h.inc)
        h.advance()
        pos_n, idx_n = oracle.generate_grid_samples(grid, n_n, step, n_grid, 0.01, h.state, h.inc)
        h.advance()
        pos = np.concatenate([pos_u, pos_n]) * np.float32(2.0) + np.float32(-1.0)
        assert np.array_equal(seen['pts'].view(np.uint32), pos.astype(np.float32).view(np.uint32))
        dt = float(bound.volume.get_diag_len()) / 64.0
        tmp = np.zeros(n_el, np.float32)
        oracle.splat_grid_samples(opacity_np(dt, pos), np.concatenate([idx_u, idx_n]), tmp)
        oracle.ema_grid_samples_nerf(tmp, grid, 0.95)
        assert np.array_equal(host(bound.density_grid).view(np.uint32), grid.view(np.uint32))
        mean = float(host(bound.get_density_grid_mean())[0])
        assert abs(mean - float(np.clip(grid, 0, None).mean())) < 1e-7
        bits = oracle.update_bitfield(grid, mean, 0.01, n_grid)
        assert np.array_equal(host(bound.density_bitfield), bits)
        step += 1
        cnt, ratio = bound.get_bitfield_count()
        assert cnt == oracle.count_bitfield(bits, n_grid) and 0 < ratio < 1
    # sampling through the refreshed bound
    rng = np.random.default_rng(0)
    o, d = _rays(rng, 256)
    near, far, mask = bound.get_near_far_from_rays({'rays_o': dev(o), 'rays_d': dev(d)})
    z, m = bound.get_zvals_from_near_far(near, far, 64, rays_o=dev(o), rays_d=dev(d))
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    dtz = np.float32(float(bound.volume.get_diag_len()) / 64)
    z_ref, m_ref, c_ref = oracle.sparse_volume_sampling_bit(o, d, host(near), host(far), 64, dtz, aabb23, n_grid, bits, 0.1,
                                                            h.state, h.inc)
    assert np.array_equal(host(m), m_ref) and np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32))
    assert c_ref.sum() > 0
    assert 'density_bitfield' in bound.state_dict() and 'density_grid' in bound.state_dict()
    bitfield_rng(reset=True)


# ---- `_multivol_func` (K11, K12, cascaded K9) ---------------------------------------------------------------------------
def _cascade_case(rng, n_grid, n_cascade, inclusive, R, frac=0.3):
    levels = n_cascade if inclusive else n_cascade - 1
    bits = np.packbits(rng.random(levels * n_grid ** 3) < frac, bitorder='little')
    o = ((rng.random((R, 3)) - 0.5) * 3.0).astype(np.float32)
    o[: R // 2] *= 0.2
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    inner = np.array([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]], np.float32)
    outer = inner * 2 ** (n_cascade - 1)
    return bits, o, d, inner, outer


@pytest.mark.parametrize('inclusive', [True, False])
@pytest.mark.parametrize('n_grid,n_cascade,n_pts,R,cone', [(16, 4, 192, 600, 1.0 / 16), (128, 5, 1024, 4096, 1.0 / 256), (32, 3, 256, 1000, 0.0)])
def test_k11_cascade_sampler_bit_exact_vs_oracle(F, oracle, inclusive, n_grid, n_cascade, n_pts, R, cone):
    rng = np.random.default_rng(n_grid + int(inclusive))
    bits, o, d, inner, outer = _cascade_case(rng, n_grid, n_cascade, inclusive, R, 0.3 if n_grid < 128 else 0.05)
    near, far, _, _ = oracle.aabb_intersection_torch(o, d, np.stack([outer[0], outer[1]], -1)[None])
    min_step = np.float32(np.sqrt(3.0) / n_pts)
    max_step = np.float32(np.sqrt(3.0) * 2 ** (n_cascade - 1) / n_grid)
    h = oracle.Pcg32(9121)
    h.advance()
    z_ref, m_ref, c_ref = oracle.sparse_sampling_in_multivol_bitfield(o, d, near, far, n_pts, cone, min_step, max_step, inner, outer,
                                                                      n_grid, n_cascade, bits, 0.05, inclusive, h.state, h.inc)
    z, m, c = F.sparse_sampling_in_multivol_bitfield(dev(o), dev(d), dev(near), dev(far), n_pts, cone, float(min_step), float(max_step),
                                                     dev(inner), dev(outer), n_grid, n_cascade, dev(bits), 0.05, inclusive, h.state,
                                                     h.inc, want_counts=True)
    assert np.array_equal(host(c), c_ref) and np.array_equal(host(m), m_ref)
    assert np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32))
    assert c_ref.sum() > 5 * R


@pytest.mark.parametrize('inclusive', [True, False])
def test_k12_and_cascaded_k9_bit_exact(F, oracle, inclusive):
    n_grid, n_cascade = 32, 5
    levels = n_cascade if inclusive else n_cascade - 1
    n = levels * n_grid ** 3 // 4
    rng = np.random.default_rng(12)
    grid = (rng.random(levels * n_grid ** 3).astype(np.float32) - 0.5)
    inner = np.array([[-0.5, -1.0, 0.0], [0.5, 1.0, 3.0]], np.float32)
    h = oracle.Pcg32(9121)
    for step, thresh in ((0, -0.01), (9, 0.3)):
        pos_ref, idx_ref = oracle.generate_grid_samples_multivol(grid, n, inner, step, n_cascade, n_grid, thresh, inclusive, h.state, h.inc)
        pos, idx = F.generate_grid_samples_multivol(dev(grid), n, dev(inner), step, n_cascade, n_grid, thresh, inclusive, h.state, h.inc)
        assert np.array_equal(host(idx), idx_ref)
        assert np.array_equal(host(pos).view(np.uint32), pos_ref.view(np.uint32))
        h.advance()
    mean = float(np.clip(grid, 0, None).mean())
    bf = torch.zeros(levels * n_grid ** 3 // 8, dtype=torch.uint8, device='cuda')
    F.update_bitfield_multivol(dev(grid), torch.tensor([mean], device='cuda'), bf, 0.01, n_grid, n_cascade, inclusive)
    assert np.array_equal(host(bf), oracle.update_bitfield_multivol(grid, mean, 0.01, n_grid, n_cascade, inclusive))
    with pytest.raises(RuntimeError):   # a grid that cannot hold every level
        F.generate_grid_samples_multivol(dev(grid[:100]), 10, dev(inner), 0, n_cascade, n_grid, 0.0, inclusive, h.state, h.inc)


def test_compat_multivol_func_module(oracle):
    import arcnerf_amd.compat as compat
    compat.install()
    import _multivol_func as M
    from arcnerf_amd.ops.multivol_func import multivol_rng
    rng = np.random.default_rng(13)
    n_grid, n_cascade, n_pts, R = 16, 3, 64, 128
    bits, o, d, inner, outer = _cascade_case(rng, n_grid, n_cascade, False, R)
    near, far, _, _ = oracle.aabb_intersection_torch(o, d, np.stack([outer[0], outer[1]], -1)[None])
    multivol_rng(reset=True)
    h = oracle.Pcg32(9121)
    z = torch.zeros((R, n_pts), device='cuda')
    m = torch.zeros((R, n_pts), dtype=torch.bool, device='cuda')
    M.sparse_sampling_in_multivol_bitfield(dev(o), dev(d), dev(near), dev(far), n_pts, 0.05, 0.03, 0.4, dev(inner), dev(outer), n_grid,
                                           n_cascade, dev(bits), 0.05, False, z, m)
    z_ref, m_ref, _ = oracle.sparse_sampling_in_multivol_bitfield(o, d, near, far, n_pts, 0.05, 0.03, 0.4, inner, outer, n_grid,
                                                                  n_cascade, bits, 0.05, False, h.state, h.inc)
    assert np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32)) and np.array_equal(host(m), m_ref)
    h.advance()
    grid = (rng.random(2 * n_grid ** 3).astype(np.float32) - 0.5)
    pos = torch.empty((500, 3), device='cuda')
    idx = torch.empty((500,), dtype=torch.int32, device='cuda')
    M.generate_grid_samples_multivol(dev(grid), 2, 500, dev(inner), n_cascade, n_grid, 0.1, False, pos, idx)
    pos_ref, idx_ref = oracle.generate_grid_samples_multivol(grid, 500, inner, 2, n_cascade, n_grid, 0.1, False, h.state, h.inc)
    assert np.array_equal(host(idx), idx_ref) and np.array_equal(host(pos).view(np.uint32), pos_ref.view(np.uint32))
    bf = torch.zeros(2 * n_grid ** 3 // 8, dtype=torch.uint8, device='cuda')
    M.update_bitfield_multivol(dev(grid), 0.2, bf, 0.01, n_grid, n_cascade, False)
    assert np.array_equal(host(bf), oracle.update_bitfield_multivol(grid, 0.2, 0.01, n_grid, n_cascade, False))
    with pytest.raises(RuntimeError):
        M.sparse_sampling_in_multivol_bitfield(dev(o), dev(d), dev(near), dev(far), n_pts, 0.05, 0.03, 0.4, dev(inner), dev(outer), n_grid,
                                               n_cascade, dev(bits[:-1]), 0.05, False, z, m)
    multivol_rng(reset=True)
