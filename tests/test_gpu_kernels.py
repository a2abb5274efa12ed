"""GPU parity tests: every HIP kernel, called through the C ABI, against (a) the golden vectors produced by the
reference's torch path and (b) the CPU oracle on seeded inputs.

Bars: integer / index / mask outputs bit-exact; sampler t values bit-exact (same IEEE op sequence, -ffp-contract=off on
both sides); fp32 outputs within 1e-5 relative (north-star bar is 1e-4 on RGB/depth).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, make_table

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from arcnerf_amd.ops import functional
    return functional


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def host(t):
    return t.detach().cpu().numpy()


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---- compositing ---------------------------------------------------------------------------------
@pytest.mark.parametrize('P', [2, 17, 64])
def test_ray_marching_vs_reference_golden(F, P):
    g = load_golden('g1_compositing')
    pre = 'P{}_'.format(P)
    sigma, rad, z = dev(g[pre + 'sigma']), dev(g[pre + 'radiance']), dev(g[pre + 'zvals'])
    for add_inf_z in (False, True):
        for mode in ('none', 'white', 'bkg_full', 'bkg_one'):
            tag = 'P{}_inf{}_{}_'.format(P, int(add_inf_z), mode)
            bkg = dev(g[pre + mode]) if mode.startswith('bkg') else None
            out = F.ray_marching_fwd(sigma, rad, z, add_inf_z=add_inf_z, white_bkg=(mode == 'white'), bkg_color=bkg)
            assert int(out['status'].item()) == 0
            for k in ('rgb', 'depth', 'mask', 'alpha', 'trans_shift', 'weights'):
                close(host(out[k]), g[tag + k], rtol=1e-5, atol=1e-6)
            d_sigma, d_rad = F.ray_marching_bwd(sigma, rad, z, dev(g[pre + 'g_rgb']), dev(g[pre + 'g_depth']),
                                                dev(g[pre + 'g_mask']), add_inf_z=add_inf_z, white_bkg=(mode == 'white'),
                                                bkg_color=bkg)
            close(host(d_rad), g[tag + 'd_radiance'], rtol=1e-5, atol=1e-6)
            close(host(d_sigma), g[tag + 'd_sigma'], rtol=2e-4, atol=2e-5)
    tag = 'P{}_alpha_'.format(P)
    out = F.ray_marching_fwd(None, rad, z, alpha=dev(g[tag + 'in']), bkg_color=dev(g[pre + 'bkg_full']))
    for k in ('rgb', 'depth', 'mask', 'trans_shift', 'weights'):
        close(host(out[k]), g[tag + k], rtol=1e-5, atol=1e-6)
    d_alpha, d_rad = F.ray_marching_bwd(None, rad, z, dev(g[pre + 'g_rgb']), dev(g[pre + 'g_depth']), dev(g[pre + 'g_mask']),
                                        alpha=dev(g[tag + 'in']), bkg_color=dev(g[pre + 'bkg_full']))
    close(host(d_rad), g[tag + 'd_radiance'], rtol=1e-5, atol=1e-6)
    close(host(d_alpha), g[tag + 'd_alpha'], rtol=2e-4, atol=2e-5)


def test_ray_marching_long_rays_vs_oracle(F, oracle):
    rng = np.random.default_rng(11)
    R, P = 300, 1024  # multi-chunk path (16 chunks of 64), ragged valid lengths with duplicated-z tails
    z = np.sort(rng.random((R, P)).astype(np.float32) * 4 + 0.5, axis=-1)
    n_valid = rng.integers(1, P + 1, size=R)
    for r in range(R):
        z[r, n_valid[r] - 1:] = z[r, n_valid[r] - 1]
    sigma = (rng.random((R, P)).astype(np.float32) ** 3) * 40
    sigma[rng.random((R, P)) < 0.5] = 0
    rad = rng.random((R, P, 3)).astype(np.float32)
    bkg = rng.random((R, 3)).astype(np.float32)
    g_rgb, g_d, g_m = (rng.normal(size=s).astype(np.float32) for s in ((R, 3), (R,), (R,)))
    for add_inf_z in (False, True):
        ref = oracle.ray_marching_fwd(sigma, rad, z, add_inf_z=add_inf_z, bkg_color=bkg)
        out = F.ray_marching_fwd(dev(sigma), dev(rad), dev(z), add_inf_z=add_inf_z, bkg_color=dev(bkg))
        for k in ('rgb', 'depth', 'mask', 'weights', 'trans_shift'):
            close(host(out[k]), ref[k], rtol=2e-5, atol=2e-6)
        rs, rr = oracle.ray_marching_bwd(sigma, rad, z, g_rgb, g_d, g_m, add_inf_z=add_inf_z, bkg_color=bkg)
        ds, dr = F.ray_marching_bwd(dev(sigma), dev(rad), dev(z), dev(g_rgb), dev(g_d), dev(g_m), add_inf_z=add_inf_z,
                                    bkg_color=dev(bkg))
        close(host(dr), rr, rtol=2e-5, atol=2e-6)
        close(host(ds), rs, rtol=5e-4, atol=5e-5)


def test_ray_marching_flags_decreasing_z(F):
    z = dev(np.array([[1.0, 0.5, 2.0]], np.float32))
    out = F.ray_marching_fwd(torch.ones(1, 3).cuda(), torch.ones(1, 3, 3).cuda(), z)
    assert int(out['status'].item()) == 1


def _packed_case(rng, R, P_dense, force_full=True):
    counts = rng.integers(0, P_dense + 1, size=R).astype(np.int32)
    if force_full:
        counts[1] = P_dense  # at least one ray as long as the dense width
    counts[0] = 0
    counts[2] = 1
    offsets = np.zeros(R + 1, np.int32)
    offsets[1:] = np.cumsum(counts)
    S = int(offsets[-1])
    t = np.zeros(S, np.float32)
    for r in range(R):
        n = counts[r]
        t[offsets[r]:offsets[r] + n] = np.sort(rng.random(n).astype(np.float32) * 3 + 0.3)
    sigma = (rng.random(S).astype(np.float32) ** 2) * 30
    rad = rng.random((S, 3)).astype(np.float32)
    return counts, offsets, t, sigma, rad


def _dense_view(counts, offsets, t, sigma, rad, P_dense):
    """the reference's padded tensors: FgModel.get_sigma_radiance_by_mask_pts (fg_model.py:305-316)"""
    R = counts.shape[0]
    z = np.zeros((R, P_dense), np.float32)
    sg = np.zeros((R, P_dense), np.float32)
    rd = np.zeros((R, P_dense, 3), np.float32)
    for r in range(R):
        n = counts[r]
        if n == 0:
            continue
        sl = slice(offsets[r], offsets[r] + n)
        z[r, :n], sg[r, :n], rd[r, :n] = t[sl], sigma[sl], rad[sl]
        z[r, n:], sg[r, n:], rd[r, n:] = t[sl][-1], sigma[sl][-1], rad[sl][-1]
    return z, sg, rd


@pytest.mark.parametrize('add_inf_z', [False, True])
@pytest.mark.parametrize('P_dense', [2, 37, 200])
def test_composite_packed_equals_dense_reference_view(F, oracle, add_inf_z, P_dense):
    rng = np.random.default_rng(100 + P_dense)
    R = 64
    counts, offsets, t, sigma, rad = _packed_case(rng, R, P_dense)
    valid = counts > 0
    z, sg, rd = _dense_view(counts, offsets, t, sigma, rad, P_dense)
    bkg = rng.random((R, 3)).astype(np.float32)
    g_rgb, g_d, g_m = (rng.normal(size=s).astype(np.float32) for s in ((R, 3), (R,), (R,)))
    ref = oracle.ray_marching_fwd(sg[valid], rd[valid], z[valid], add_inf_z=add_inf_z, bkg_color=bkg[valid])
    pd = torch.tensor([P_dense], dtype=torch.int32).cuda()
    out = F.composite_packed_fwd(dev(sigma), dev(rad), dev(t), dev(offsets), p_dense=2, p_dense_dev=pd, add_inf_z=add_inf_z,
                                 bkg_color=dev(bkg))
    for k in ('rgb', 'depth', 'mask'):
        close(host(out[k])[valid], ref[k], rtol=2e-5, atol=2e-6)
    # rays without samples: rgb = bkg (T=1), depth = mask = 0
    close(host(out['rgb'])[~valid], bkg[~valid], rtol=0, atol=0)
    assert (host(out['mask'])[~valid] == 0).all()
    rs, rr = oracle.ray_marching_bwd(sg[valid], rd[valid], z[valid], g_rgb[valid], g_d[valid], g_m[valid],
                                     add_inf_z=add_inf_z, bkg_color=bkg[valid])
    ds, dr = F.composite_packed_bwd(dev(sigma), dev(rad), dev(t), dev(offsets), dev(g_rgb), dev(g_d), dev(g_m), p_dense=2,
                                    p_dense_dev=pd, add_inf_z=add_inf_z, bkg_color=dev(bkg))
    ds, dr = host(ds), host(dr)
    # scatter the dense reference gradients back onto packed samples: valid columns map 1:1, the duplicated tail
    # columns alias the last valid sample (index_put of sigma[mask_pts] = _sigma is 1:1, the `last_sigma` fill is a
    # gather of the last packed sample: its gradient flows to that sample)
    want_s = np.zeros_like(sigma)
    want_r = np.zeros_like(rad)
    vi = np.nonzero(valid)[0]
    for k, r in enumerate(vi):
        n = counts[r]
        want_s[offsets[r]:offsets[r] + n] = rs[k, :n]
        want_r[offsets[r]:offsets[r] + n] = rr[k, :n]
        want_s[offsets[r] + n - 1] += rs[k, n:].sum()
        want_r[offsets[r] + n - 1] += rr[k, n:].sum(0)
    close(dr, want_r, rtol=2e-5, atol=2e-6)
    close(ds, want_s, rtol=5e-4, atol=5e-5)


# ---- resampling ----------------------------------------------------------------------------------
def test_sample_cdf_vs_reference_golden(F):
    g = load_golden('g2_resampling')
    s, inds = F.sample_cdf(dev(g['bins']), dev(g['cdf']), dev(g['u_det']), want_inds=True)
    assert (host(inds) == g['inds_det']).all()
    close(host(s), g['samples_det'], rtol=1e-6, atol=1e-6)
    s, inds = F.sample_cdf(dev(g['bins']), dev(g['cdf']), dev(g['u']), want_inds=True)
    assert (host(inds) == g['inds_rnd']).all()
    close(host(s), g['samples_rnd'], rtol=1e-6, atol=1e-6)


# ---- bounds --------------------------------------------------------------------------------------
def test_aabb_torch_semantics_vs_reference_golden(F):
    g = load_golden('g4_intersections')
    near, far, pts, mask = F.aabb_intersection_torch(dev(g['rays_o']), dev(g['rays_d']), dev(g['aabb']))
    assert (host(mask) == g['mask']).all()
    close(host(near), g['near'], rtol=2e-6, atol=2e-6)
    close(host(far), g['far'], rtol=2e-6, atol=2e-6)
    close(host(pts), g['pts'], rtol=1e-5, atol=1e-5)


def test_sphere_intersection_vs_reference_golden_and_oracle(F, oracle):
    g = load_golden('g4_intersections')
    near, far, pts, mask = F.sphere_intersection(dev(g['rays_o']), dev(g['rays_d']), dev(g['radius']))
    assert (host(mask) == g['s_mask']).all()
    close(host(near), g['s_near'], rtol=1e-6, atol=1e-6)
    close(host(far), g['s_far'], rtol=1e-6, atol=1e-6)
    close(host(pts), g['s_pts'], rtol=1e-6, atol=2e-6)
    # bit-exact against the C restatement (same operation order), shifted origin and a scalar radius included
    rng = np.random.default_rng(12)
    o = (rng.normal(size=(5000, 3)) * 1.5).astype(np.float32)
    d = rng.normal(size=(5000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    org = (0.25, -0.5, 0.125)
    ref = oracle.sphere_intersection(o, d, 1.3, org)
    got = F.sphere_intersection(dev(o), dev(d), 1.3, org)
    for a, b in zip(got, ref):
        assert np.array_equal(host(a), b)
    # the module mirror: Sphere / SphereBound
    from arcnerf_amd.models.base_modules.obj_bound import build_obj_bound
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    bound, kind = build_obj_bound(dict_to_obj({'obj_bound': {'sphere': {'origin': list(org), 'radius': 1.3}}}))
    bound = bound.cuda()
    nr, fr, mask_rays = bound.get_near_far_from_rays({'rays_o': dev(o), 'rays_d': dev(d)})
    assert kind == 'sphere' and nr.shape == (5000, 1) and mask_rays.shape == (5000,)
    assert np.array_equal(host(nr), ref[0]) and np.array_equal(host(mask_rays), ref[3][:, 0])
    zvals, mask_pts = bound.get_zvals_from_near_far(nr, fr, 16, inference_only=True)
    assert zvals.shape == (5000, 16) and mask_pts is None


def test_k2_bit_exact_vs_oracle(F, oracle):
    g = load_golden('g4_intersections')
    aabb23 = np.ascontiguousarray(np.transpose(g['aabb'], (0, 2, 1)))
    ref = oracle.aabb_intersection(g['rays_o'], g['rays_d'], aabb23)
    got = F.aabb_intersection(dev(g['rays_o']), dev(g['rays_d']), dev(aabb23))
    for a, b in zip(got, ref):
        assert np.array_equal(host(a), b)


def test_k1_vs_reference_golden_and_oracle(F, oracle):
    g = load_golden('g5_voxel')
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    occ = F.check_pts_in_occ_voxel(dev(g['pts']), dev(g['n8_bitfield']), dev(aabb23), 8)
    assert (host(occ) == g['n8_pts_in_occ']).all()
    rng = np.random.default_rng(3)
    bf = rng.random((128, 128, 128)) < 0.1
    pts = (rng.random((200000, 3)).astype(np.float32) - 0.5) * 2.2
    assert np.array_equal(host(F.check_pts_in_occ_voxel(dev(pts), dev(bf), dev(aabb23), 128)),
                          oracle.check_pts_in_occ_voxel(pts, bf, aabb23, 128))


# ---- sampler -------------------------------------------------------------------------------------
def _rays(rng, R, radius=2.86):
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = (o / np.linalg.norm(o, axis=-1, keepdims=True) * radius).astype(np.float32)
    tgt = (rng.random((R, 3)).astype(np.float32) - 0.5) * 1.8
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return o, d


def _blob_bitfield(n_grid, rng, frac=0.06):
    """a few boxes/spheres filling about `frac` of the grid"""
    ax = (np.arange(n_grid) + 0.5) / n_grid * 2 - 1
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    bf = np.zeros((n_grid,) * 3, bool)
    while bf.mean() < frac:
        c = (rng.random(3) - 0.5) * 1.2
        r = rng.random() * 0.25 + 0.1
        if rng.random() < 0.5:
            bf |= ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) < r * r
        else:
            bf |= (np.abs(X - c[0]) < r) & (np.abs(Y - c[1]) < r * 0.7) & (np.abs(Z - c[2]) < r * 0.5)
    return bf


@pytest.mark.parametrize('n_grid,n_pts,R', [(16, 256, 500), (128, 1024, 4096)])
def test_k3_sampler_bit_exact_vs_oracle(F, oracle, n_grid, n_pts, R):
    rng = np.random.default_rng(n_grid)
    bf = _blob_bitfield(n_grid, rng, 0.15 if n_grid == 16 else 0.05)
    o, d = _rays(rng, R)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, _ = oracle.aabb_intersection(o, d, aabb23[None])
    dt = np.float32(2 * np.sqrt(3.0) / n_pts)
    hostrng = oracle.Pcg32(9121)
    hostrng.advance()  # second launch of the process: state after one 2^32 jump
    z_ref, m_ref, c_ref = oracle.sparse_volume_sampling(o, d, near, far, n_pts, dt, aabb23, n_grid, bf, 0.2, hostrng.state,
                                                        hostrng.inc)
    gr = F.Pcg32Host(9121)
    gr.advance()
    assert (gr.state, gr.inc) == (hostrng.state, hostrng.inc)
    z, m, c = F.sparse_volume_sampling(dev(o), dev(d), dev(near), dev(far), n_pts, float(dt), dev(aabb23), n_grid, dev(bf), 0.2,
                                       gr.state, gr.inc, want_counts=True)
    assert np.array_equal(host(c), c_ref)
    assert np.array_equal(host(m), m_ref)
    assert np.array_equal(host(z).view(np.uint32), z_ref.view(np.uint32))  # t values: 0 ulp
    assert c_ref.sum() > 10 * R / 4

    # compacted form == boolean-mask compaction of the dense form (row-major), K2 bounds fused
    pk = F.march_packed(dev(o), dev(d), dev(aabb23), n_grid, dev(bf), n_pts, float(dt), 0.2, gr.state, gr.inc)
    total = int(pk['offsets'][-1].item())
    assert total == int(c_ref.sum())
    assert np.array_equal(host(pk['counts']), c_ref)
    assert np.array_equal(host(pk['t'][:total]).view(np.uint32), z_ref[m_ref].view(np.uint32))
    rid = np.repeat(np.arange(R, dtype=np.int32), c_ref)
    assert np.array_equal(host(pk['ray_id'][:total]), rid)
    assert np.array_equal(host(pk['near']), near[:, 0]) and np.array_equal(host(pk['far']), far[:, 0])
    # packed 1-bit occupancy gives the identical result
    bits = np.packbits(bf.reshape(-1), bitorder='little')
    pk2 = F.march_packed(dev(o), dev(d), dev(aabb23), n_grid, dev(bits), n_pts, float(dt), 0.2, gr.state, gr.inc,
                         packed_bits=True)
    assert np.array_equal(host(pk2['offsets']), host(pk['offsets']))
    assert np.array_equal(host(pk2['t'][:total]).view(np.uint32), host(pk['t'][:total]).view(np.uint32))
    # points of packed samples
    xyz, dirs = F.packed_points(dev(o), dev(d), pk['t'], pk['ray_id'], n=total)
    want = o[rid] + z_ref[m_ref][:, None] * d[rid]
    assert np.array_equal(host(xyz), want.astype(np.float32))
    assert np.array_equal(host(dirs), d[rid])


def test_k4_reduce_max_vs_oracle(F, oracle):
    rng = np.random.default_rng(4)
    n, ng = 100000, 5000
    full = rng.random(n).astype(np.float32)
    idx = rng.integers(0, ng, size=n)
    assert np.array_equal(host(F.tensor_reduce_max(dev(full), dev(idx), ng)), oracle.tensor_reduce_max(full, idx, ng))


# ---- hash grid -----------------------------------------------------------------------------------
def _desc(F_, res, offs, n_feat):
    from arcnerf_amd import _native as N
    return N.make_hashgrid_desc(res, offs, n_feat, [-1, -1, -1], [1, 1, 1])


@pytest.mark.parametrize('tag', ['ngp', 'tiny', 'f4'])
def test_hashgrid_vs_reference_golden(F, oracle, tag):
    g = load_golden('g6_hashgrid')
    L, nf, T, base, mx_res = [int(v) for v in g[tag + '_cfg']]
    res, offs = g[tag + '_resolutions'], g[tag + '_offsets']
    table = make_table(int(offs[-1]), nf, seed=7, scale=0.5)
    desc = _desc(F, res, offs, nf)
    tb = dev(table)
    out, idx = F.hashgrid_fwd(dev(g[tag + '_xyz']), tb, desc, want_idx=True)
    assert np.array_equal(host(idx), g[tag + '_hash_idx'])  # integer hash rows: bit exact vs the reference
    close(host(out), g[tag + '_out'], rtol=1e-5, atol=1e-6)
    dtable, dxyz = F.hashgrid_bwd(dev(g[tag + '_xyz']), tb, dev(g[tag + '_g_out']), desc, want_dxyz=True)
    dtable = host(dtable)
    rows = g[tag + '_d_table_rows']
    nz = np.nonzero(np.abs(dtable).sum(-1) > 0)[0]
    assert set(nz.tolist()) <= set(rows.tolist())
    close(dtable[rows], g[tag + '_d_table_vals'], rtol=1e-4, atol=1e-5)
    scale = np.abs(g[tag + '_d_xyz']).max()
    close(host(dxyz), g[tag + '_d_xyz'], rtol=1e-4, atol=1e-4 * scale)


def test_level_major_variants_match_row_major(F):
    """arcn_hashgrid_fwd_xcd (row- and level-major), arcn_mlp_fwd_lm / arcn_mlp_bwd_lm and arcn_hashgrid_bwd_lm against the
    row-major entry points on the same inputs: gathers bit-identical, GEMM / scatter results within summation-order noise."""
    import ctypes as C
    from arcnerf_amd import _native as N
    from arcnerf_amd.pipeline import hashgrid_level_table
    rng = np.random.default_rng(21)
    res, offs = hashgrid_level_table(16, 19, 16, 2048)
    desc = _desc(F, res, offs, 2)
    table = dev(make_table(int(offs[-1]), 2, seed=3, scale=0.5))
    S, cap = 9001, 9216   # ragged count, level stride = capacity
    xyz = dev(((rng.random((S, 3)).astype(np.float32) - 0.5) * 2.05).astype(np.float32))
    lib, st = N.lib(), N.stream()
    ref = F.hashgrid_fwd_plain(xyz, table, desc)
    rm = torch.zeros(S, 32, device='cuda')
    lm = torch.zeros(16, cap, 2, device='cuda')
    N.check(lib.arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(rm), 0, S, S, None, st))
    N.check(lib.arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, cap, S, None, st))
    assert torch.equal(rm, ref)
    assert torch.equal(lm[:, :S].permute(1, 0, 2).reshape(S, 32), ref)
    # geometry net on the level-major features
    dims = [32, 64, 16]
    mdesc = N.make_mlp_desc(dims, 'relu', None)
    w = dev((rng.normal(size=32 * 64 + 64 * 16) * 0.2).astype(np.float32))
    out_ref, acts_ref = F.mlp_fwd(ref, w, None, mdesc, save_acts=True)
    out = torch.zeros(S, 16, device='cuda')
    acts = torch.zeros(F.mlp_acts_floats(mdesc, cap), device='cuda')
    N.check(lib.arcn_mlp_fwd_lm(N.ptr(lm), cap, N.ptr(w), C.addressof(mdesc), N.ptr(out), N.ptr(acts), cap, S, None, st))
    close(host(out), host(out_ref), rtol=1e-6, atol=1e-6)
    dout = dev(rng.normal(size=(S, 16)).astype(np.float32))
    dx_ref, dw_ref, _ = F.mlp_bwd(ref, w, None, mdesc, out_ref, acts_ref, dout)
    dx_lm = torch.zeros(16, cap, 2, device='cuda')
    dw = torch.zeros_like(w)
    scr = torch.zeros(F.mlp_scratch_floats(mdesc, cap), device='cuda')
    N.check(lib.arcn_mlp_bwd_lm(N.ptr(lm), cap, N.ptr(w), C.addressof(mdesc), N.ptr(out), N.ptr(acts), N.ptr(dout), N.ptr(dx_lm),
                                N.ptr(dw), N.ptr(scr), 0, cap, S, None, st))
    close(host(dx_lm[:, :S].permute(1, 0, 2).reshape(S, 32)), host(dx_ref), rtol=1e-5, atol=1e-6)
    close(host(dw), host(dw_ref), rtol=1e-4, atol=1e-4)
    # scatter of the level-major gradient
    dt_ref, _ = F.hashgrid_bwd(xyz, table, dx_ref, desc, workspace=True)
    dt = torch.zeros_like(table)
    ws = F.hashgrid_bwd_workspace(desc, S, 'cuda')
    N.check(lib.arcn_hashgrid_bwd_lm(N.ptr(xyz), N.ptr(dx_lm), cap, C.addressof(desc), N.ptr(dt), N.ptr(ws), ws.numel(), S, None, st))
    close(host(dt), host(dt_ref), rtol=1e-4, atol=1e-5)
    # shapes the level-major path is not wired for are argument errors
    bad = N.make_mlp_desc([32, 64, 64, 3], 'relu', 'sigmoid')
    assert lib.arcn_mlp_fwd_lm(N.ptr(lm), cap, N.ptr(w), C.addressof(bad), N.ptr(out), None, cap, S, None, st) == -1


def test_glue_with_per_ray_harmonics_is_bit_identical(F):
    import ctypes as C
    from arcnerf_amd import _native as N
    rng = np.random.default_rng(4)
    R, S = 300, 7000
    rays_d = dev((rng.normal(size=(R, 3)) * rng.uniform(0.5, 2.0, size=(R, 1))).astype(np.float32))
    ray_id = torch.from_numpy(np.sort(rng.integers(0, R, size=S)).astype(np.int32)).cuda()
    dirs = rays_d[ray_id.long()].contiguous()
    geo = dev(rng.normal(size=(S, 16)).astype(np.float32))
    for feat_first in (True, False):
        ref_in, ref_sigma = F.ngp_glue_fwd(geo, dirs, 0, 16, 4, feat_first=feat_first, sigma_act='truncexp')
        sh_ray = torch.zeros(R, 16, device='cuda')
        rad_in = torch.zeros(S, 32, device='cuda')
        sigma = torch.zeros(S, device='cuda')
        lib, st = N.lib(), N.stream()
        N.check(lib.arcn_ngp_ray_sh(N.ptr(rays_d), 4, N.ptr(sh_ray), R, st))
        N.check(lib.arcn_ngp_glue_fwd_rays(N.ptr(geo), N.ptr(sh_ray), N.ptr(ray_id), 16, 0, 16, 4, int(feat_first), N.ACT['truncexp'],
                                           N.ptr(rad_in), N.ptr(sigma), S, None, st))
        assert torch.equal(rad_in, ref_in) and torch.equal(sigma, ref_sigma)
        # the radiance net with the glue folded into its operand load (fwd + fused bwd) against glue kernel + plain net
        for dims, act_out in (([32, 64, 64, 3], 'sigmoid'), ([32, 64, 16], None)):
            mdesc = N.make_mlp_desc(dims, 'relu', act_out)
            w = dev((rng.normal(size=sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))) * 0.2).astype(np.float32))
            out_ref, acts_ref = F.mlp_fwd(ref_in, w, None, mdesc, save_acts=True)
            out = torch.zeros(S, dims[-1], device='cuda')
            acts = torch.zeros(F.mlp_acts_floats(mdesc, S), device='cuda')
            sig2 = torch.zeros(S, device='cuda')
            N.check(lib.arcn_mlp_fwd_cat(N.ptr(geo), N.ptr(sh_ray), N.ptr(ray_id), int(feat_first), N.ptr(w), C.addressof(mdesc),
                                         N.ptr(out), N.ptr(acts), N.ptr(sig2), N.ACT['truncexp'], S, S, None, st))
            assert torch.equal(out, out_ref) and torch.equal(sig2, ref_sigma)
            dout = dev(rng.normal(size=(S, dims[-1])).astype(np.float32))
            d_sigma = dev(rng.normal(size=S).astype(np.float32))
            dx_ref, dw_ref, _ = F.mlp_bwd(ref_in, w, None, mdesc, out_ref, acts_ref, dout)
            dgeo_ref = F.ngp_glue_bwd(geo, dx_ref, d_sigma, 0, 16, 4, feat_first=feat_first, sigma_act='truncexp')
            dgeo = torch.zeros(S, 16, device='cuda')
            dw = torch.zeros_like(w)
            scr = torch.zeros(F.mlp_scratch_floats(mdesc, S), device='cuda')
            N.check(lib.arcn_mlp_bwd_cat(N.ptr(geo), N.ptr(sh_ray), N.ptr(ray_id), int(feat_first), N.ptr(w), C.addressof(mdesc),
                                         N.ptr(out), N.ptr(acts), N.ptr(dout), N.ptr(dgeo), N.ptr(d_sigma), N.ACT['truncexp'],
                                         N.ptr(dw), N.ptr(scr), 1, S, S, None, st))
            assert float(dw.abs().max()) == 0.0                      # deferred: the partials are still in the scratch
            N.check(lib.arcn_mlp_bwd_reduce(C.addressof(mdesc), N.ptr(scr), N.ptr(dw), S, S, st))
            close(host(dgeo), host(dgeo_ref), rtol=1e-5, atol=1e-6)
            close(host(dw), host(dw_ref), rtol=1e-5, atol=1e-5)


def test_hashgrid_scatter_bin_overflow_and_runs(F, oracle):
    """Adversarial sample distributions for the binned scatter: (a) samples alternating between two far-apart cells (runs of
    length 1, every record of a hashed level lands in the same <= 8 bins: their fixed capacity overflows and the direct-atomic
    path must take the excess), (b) all samples inside one cell (one run per wave: the segmented reduction carries everything)."""
    rng = np.random.default_rng(8)
    res, offs = oracle.hashgrid_levels(16, 19, 16, 2048)
    table = make_table(int(offs[-1]), 2, seed=2, scale=0.5)
    desc = _desc(F, res, offs, 2)
    mn, mx = np.full(3, -1, np.float32), np.full(3, 1, np.float32)
    S = 40000
    a = np.array([0.3137, -0.2711, 0.5519], np.float32)
    b = np.array([-0.6123, 0.4401, -0.1907], np.float32)
    jit = (rng.random((S, 3)).astype(np.float32) - 0.5) * np.float32(2e-4)   # stays inside one finest-level cell (1e-3 wide)
    for name, xyz in (('alternating', np.where((np.arange(S) % 2 == 0)[:, None], a, b) + jit), ('one cell', a + jit)):
        xyz = xyz.astype(np.float32)
        gout = rng.normal(size=(S, 32)).astype(np.float32)
        ref_dt = oracle.hashgrid_bwd(xyz, table, gout, res, offs, mn, mx)
        dt, _ = F.hashgrid_bwd(dev(xyz), dev(table), dev(gout), desc, workspace=True)
        scale = np.abs(ref_dt).max()
        assert np.abs(host(dt) - ref_dt).max() < 2e-5 * scale, name   # sums of 2e4 terms per row: fp32 summation order noise


def test_hashgrid_ngp_large_vs_oracle(F, oracle):
    rng = np.random.default_rng(6)
    res, offs = oracle.hashgrid_levels(16, 19, 16, 2048)
    table = make_table(int(offs[-1]), 2, seed=9, scale=0.5)
    desc = _desc(F, res, offs, 2)
    S = 20000
    xyz = ((rng.random((S, 3)).astype(np.float32) - 0.5) * 2.05).astype(np.float32)
    mn, mx = np.full(3, -1, np.float32), np.full(3, 1, np.float32)
    ref, ref_idx = oracle.hashgrid_fwd(xyz, table, res, offs, mn, mx, with_idx=True)
    tb = dev(table)
    out, idx = F.hashgrid_fwd(dev(xyz), tb, desc, want_idx=True)
    assert np.array_equal(host(idx).astype(np.int64), ref_idx)
    close(host(out), ref, rtol=1e-6, atol=1e-7)
    gout = rng.normal(size=ref.shape).astype(np.float32)
    ref_dt = oracle.hashgrid_bwd(xyz, table, gout, res, offs, mn, mx)
    dt, _ = F.hashgrid_bwd(dev(xyz), tb, dev(gout), desc)
    close(host(dt), ref_dt, rtol=1e-4, atol=1e-5)
    # owner-computes scatter through LDS (no global atomics on the large levels): same result, and it ADDS into dtable
    ws = True
    dt2, _ = F.hashgrid_bwd(dev(xyz), tb, dev(gout), desc, workspace=ws)
    close(host(dt2), ref_dt, rtol=1e-4, atol=1e-5)
    F.hashgrid_bwd(dev(xyz), tb, dev(gout), desc, workspace=ws, dtable=dt2)
    close(host(dt2), 2 * ref_dt, rtol=1e-4, atol=2e-5)
    # points exactly on / next to voxel faces exercise the exact-division fallback of the fast cell index
    edge = np.round(xyz[:4000] * 64) / 64
    edge[::3] += np.float32(1e-7)
    edge = edge.astype(np.float32)
    g2 = gout[:4000]
    ref_e = oracle.hashgrid_bwd(edge, table, g2, res, offs, mn, mx)
    dte, _ = F.hashgrid_bwd(dev(edge), tb, dev(g2), desc, workspace=ws)
    close(host(dte), ref_e, rtol=1e-4, atol=1e-5)
    # device-side sample count: only the first n_dev samples are touched
    n_dev = torch.tensor([1234], dtype=torch.int32).cuda()
    out2 = torch.full((S, 32), 7.0, device='cuda')
    F.hashgrid_fwd(dev(xyz), tb, desc, n_dev=n_dev, out=out2)
    o2 = host(out2)
    assert np.array_equal(o2[:1234], host(out)[:1234]) and (o2[1234:] == 7.0).all()


# ---- freq / SH -----------------------------------------------------------------------------------
def test_freq_sh_vs_reference_golden(F):
    g = load_golden('g7_freq_sh')
    for n_freqs in (10, 4, 0):
        for inc in (True, False):
            if n_freqs == 0 and not inc:
                continue
            t = 'freq{}_inc{}'.format(n_freqs, int(inc))
            close(host(F.freq_fwd(dev(g['x']), n_freqs, inc)), g[t], rtol=0, atol=5e-6)
            close(host(F.freq_bwd(dev(g['x']), dev(g[t + '_g']), n_freqs, inc)), g[t + '_dx'], rtol=1e-5, atol=3e-3)
    for deg in (1, 2, 3, 4, 5):
        for inc in (True, False):
            close(host(F.sh_fwd(dev(g['dirs']), deg, inc)), g['sh{}_inc{}'.format(deg, int(inc))], rtol=1e-6, atol=1e-6)
    y = F.act_fwd(dev(g_truncexp()[0]), 'truncexp')
    close(host(y), g_truncexp()[1], rtol=2e-6, atol=0)
    close(host(F.act_bwd(dev(g_truncexp()[0]), y, torch.ones_like(y), 'truncexp')), g_truncexp()[2], rtol=2e-6, atol=0)


def g_truncexp():
    g = load_golden('g8_mlps')
    return g['truncexp_x'], g['truncexp_y'], g['truncexp_dx']


# ---- fused MLP -----------------------------------------------------------------------------------
def _layers(g, prefix):
    ws = sorted([k for k in g.files if k.startswith(prefix + '.layers.') and k.endswith('weight')],
                key=lambda k: int(k[len(prefix) + 8:].split('.')[0]))
    return [(g[k], g[k[:-6] + 'bias'] if (k[:-6] + 'bias') in g.files else None) for k in ws]


def _flat(layers):
    w = np.concatenate([W.reshape(-1) for W, _ in layers]).astype(np.float32)
    b = None if layers[0][1] is None else np.concatenate([b for _, b in layers]).astype(np.float32)
    return w, b


@pytest.mark.parametrize('bias', [0, 1])
def test_fused_mlp_vs_reference_golden(F, bias):
    """NGP-shaped geo (32->64->16) and radiance (32->64->64->3) nets against the reference's torch GeoNet/RadianceNet."""
    from arcnerf_amd import _native as N
    g = load_golden('g8_mlps')
    t = 'geo_b{}'.format(bias)
    layers = _layers(g, t)
    w, b = _flat(layers)
    desc = N.make_mlp_desc([32, 64, 16], 'relu', None, has_bias=bool(bias))
    x = dev(g[t + '_x'])
    wd, bd = dev(w), (dev(b) if bias else None)
    out, acts = F.mlp_fwd(x, wd, bd, desc, save_acts=True)
    o = host(out)
    close(np.exp(o[:, :1]), g[t + '_sigma'], rtol=2e-5, atol=1e-6)
    close(o[:, 1:], g[t + '_feat'], rtol=1e-5, atol=2e-6)
    # backward: d out = [g_sigma * exp(clamp(o0)), g_feat]
    do = np.concatenate([g[t + '_g_sigma'] * np.exp(np.clip(o[:, :1], -15, 15)), g[t + '_g_feat']], 1).astype(np.float32)
    dx, dw, db = F.mlp_bwd(x, wd, bd, desc, out, acts, dev(do))
    names = sorted(k for k in g.files if k.startswith(t + '_grad.') and k.endswith('weight'))
    want_w = np.concatenate([g[k].reshape(-1) for k in names])
    close(host(dw), want_w, rtol=1e-4, atol=1e-4)
    close(host(dx), g[t + '_dx'], rtol=1e-4, atol=1e-5)
    if bias:
        bn = sorted(k for k in g.files if k.startswith(t + '_grad.') and k.endswith('bias'))
        close(host(db), np.concatenate([g[k] for k in bn]), rtol=1e-4, atol=1e-4)

    t = 'rad_b{}'.format(bias)
    layers = _layers(g, t)
    w, b = _flat(layers)
    desc = N.make_mlp_desc([32, 64, 64, 3], 'relu', 'sigmoid', has_bias=bool(bias))
    v = g[t + '_view']
    v = (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)
    xin = torch.cat([dev(g[t + '_feat']), F.sh_fwd(dev(v), 4, False)], 1).contiguous()  # mode 'fv'
    wd, bd = dev(w), (dev(b) if bias else None)
    rgb, acts = F.mlp_fwd(xin, wd, bd, desc, save_acts=True)
    close(host(rgb), g[t + '_rgb'], rtol=1e-5, atol=2e-6)
    dx, dw, db = F.mlp_bwd(xin, wd, bd, desc, rgb, acts, dev(g[t + '_g_rgb']))
    names = sorted(k for k in g.files if k.startswith(t + '_grad.') and k.endswith('weight'))
    close(host(dw), np.concatenate([g[k].reshape(-1) for k in names]), rtol=1e-4, atol=1e-4)
    close(host(dx)[:, :16], g[t + '_dfeat'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('dims,act_out,bias,S', [([32, 64, 16], None, False, 5000), ([32, 64, 64, 3], 'sigmoid', False, 70001),
                                                 ([27, 48, 5], 'sigmoid', True, 333), ([63, 128, 128, 17], None, True, 1000),
                                                 ([16, 16], 'relu', False, 64),
                                                 # the other tile shapes of the fused dx + dW backward (64-wide input)
                                                 ([64, 64, 16], None, False, 4097), ([50, 64, 64, 16], 'sigmoid', False, 3000)])
def test_fused_mlp_shapes_vs_oracle(F, oracle, dims, act_out, bias, S):
    from arcnerf_amd import _native as N
    rng = np.random.default_rng(sum(dims))
    Ws = [(rng.normal(size=(dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.normal(size=dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)] if bias else None
    x = rng.normal(size=(S, dims[0])).astype(np.float32)

    def fwd(x):
        hs, pres = [x], []
        for i, W in enumerate(Ws):
            last = i == len(Ws) - 1
            y, pre = oracle.linear_fwd(hs[-1], W, bs[i] if bias else None, act_out if last else 'relu', want_pre=True)
            hs.append(y)
            pres.append(pre)
        return hs, pres

    # a ReLU pre-activation within rounding noise of 0 may change sign with the summation order (MFMA k-order vs the
    # oracle's): the derivative is discontinuous there, so such samples are removed from the comparison set
    hs, pres = fwd(x)
    relu_pres = pres if act_out == 'relu' else pres[:-1]
    if relu_pres:
        keep = np.ones(S, bool)
        for p_ in relu_pres:
            keep &= (np.abs(p_) > 1e-5).all(-1)
        x = np.ascontiguousarray(x[keep])
        S = x.shape[0]
        hs, pres = fwd(x)
    desc = N.make_mlp_desc(dims, 'relu', act_out, has_bias=bias)
    w = dev(np.concatenate([W.reshape(-1) for W in Ws]))
    b = dev(np.concatenate(bs)) if bias else None
    xd = dev(x)
    out, acts = F.mlp_fwd(xd, w, b, desc, save_acts=True)
    close(host(out), hs[-1], rtol=2e-5, atol=2e-6)
    dout = rng.normal(size=hs[-1].shape).astype(np.float32)
    dy = dout
    gW, gb = [], []
    for i in reversed(range(len(Ws))):
        last = i == len(Ws) - 1
        dy, dW, db = oracle.linear_bwd(hs[i], Ws[i], pres[i], hs[i + 1], dy, act_out if last else 'relu', has_bias=bias)
        gW.insert(0, dW)
        gb.insert(0, db)
    dx, dw, dbb = F.mlp_bwd(xd, w, b, desc, out, acts, dev(dout))
    close(host(dx), dy, rtol=1e-4, atol=1e-5)
    ref_w = np.concatenate([m.reshape(-1) for m in gW])
    close(host(dw), ref_w, rtol=2e-4, atol=2e-4 * np.abs(ref_w).max())
    if bias:
        ref_b = np.concatenate(gb)
        close(host(dbb), ref_b, rtol=2e-4, atol=2e-4 * np.abs(ref_b).max())


# ---- occupancy update / optimiser ----------------------------------------------------------------
def test_occupancy_update_vs_reference_golden(F):
    g = load_golden('g10_occupancy')
    opa = dev(g['opa0'].reshape(-1).copy())
    F.update_opafield(opa, dev(g['flat_idx'].astype(np.int64)), dev(g['new_opacity']), ema=0.95)
    assert np.array_equal(host(opa).reshape(8, 8, 8), g['opa1'])
    bf = torch.zeros(512, dtype=torch.bool, device='cuda')
    F.update_bitfield_by_opafield(opa, bf, 0.01)
    assert np.array_equal(host(bf).reshape(8, 8, 8), g['bitfield'])


def test_adam_ema_vs_torch(F):
    """torch.optim.Adam + the reference's EMA.ema_step formula (arcnerf/trainer/ema.py:29-43) in plain torch fp32."""
    torch.manual_seed(0)
    n = 100003
    decay = 0.95
    p0 = torch.randn(n)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-1, eps=1e-15, weight_decay=1e-6)
    p = p0.clone().cuda()
    m, v, ema = torch.zeros(n).cuda(), torch.zeros(n).cuda(), p0.clone().cuda()
    old_avg = p0.clone()
    for step in range(1, 6):
        gr = torch.randn(n) * (10.0 ** torch.randint(-6, 1, (n,)).float())
        p_ref.grad = gr.clone()
        opt.step()
        with torch.no_grad():
            new_avg = ((1 - decay) * p_ref + decay * old_avg * (1 - decay ** (step - 1))) * (1.0 / (1 - decay ** step))
            p_ref.copy_(new_avg)
            old_avg = new_avg.clone()
        gd = gr.cuda()
        F.adam_ema_step(p, gd, m, v, ema, step, lr=1e-1, eps=1e-15, weight_decay=1e-6, ema_decay=decay, zero_grad=True)
        assert float(gd.abs().max()) == 0.0
        close(host(p), p_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
        close(host(ema), old_avg.numpy(), rtol=2e-5, atol=2e-6)
    # without EMA: plain Adam
    q_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([q_ref], lr=1e-2)
    q, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 4):
        gr = torch.randn(n)
        q_ref.grad = gr.clone()
        opt.step()
        F.adam_ema_step(q, gr.cuda(), m, v, None, step, lr=1e-2)
        close(host(q), q_ref.detach().numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('tag,s', [('s64', 64.0), ('s800', 800.0)])
@pytest.mark.parametrize('clip', [True, False])
def test_sdf_to_alpha_vs_reference_golden_and_oracle(F, oracle, tag, s, clip):
    from arcnerf_amd.models.neus_model import sdf_to_alpha, sdf_to_cdf, sdf_to_pdf
    g = load_golden('g12_neus')
    key = '{}_clip{}'.format(tag, int(clip))
    sd = dev(g['mid_sdf']).requires_grad_(True)
    sl = dev(g['mid_slope']).requires_grad_(True)
    sv = torch.tensor(s, device='cuda', requires_grad=True)
    alpha = sdf_to_alpha(sd, dev(g['zvals']), sl, sv, clip=clip)
    close(host(alpha.detach()), g[key + '_alpha'], rtol=1e-6, atol=5e-7)
    (alpha * dev(g[key + '_gout'])).sum().backward()
    close(host(sd.grad), g[key + '_d_sdf'], rtol=1e-4, atol=1e-5 * np.abs(g[key + '_d_sdf']).max())
    close(host(sl.grad), g[key + '_d_slope'], rtol=1e-4, atol=1e-5 * np.abs(g[key + '_d_slope']).max())
    assert abs(float(sv.grad) - float(g[key + '_d_s'])) <= 2e-4 * abs(float(g[key + '_d_s'])) + 1e-6
    close(host(sdf_to_cdf(sd.detach(), s)), g[tag + '_cdf'], rtol=1e-6, atol=1e-7)
    pdf, ref_pdf = host(sdf_to_pdf(sd.detach(), s)), g[tag + '_pdf']
    fin = np.isfinite(ref_pdf)                       # the reference's formula overflows to inf/inf for s * |sdf| > 88
    assert np.array_equal(np.isfinite(pdf), fin)
    close(pdf[fin], ref_pdf[fin], rtol=1e-5, atol=1e-6 * np.abs(ref_pdf[fin]).max())
    # larger, ragged problem against the C restatement
    rng = np.random.default_rng(5)
    R, P = 3001, 129
    z = np.sort(rng.random((R, P)).astype(np.float32) * 3 + 0.2, axis=-1)
    msd = ((rng.random((R, P - 1)) - 0.4) * 0.5).astype(np.float32)
    msl = (-rng.random((R, P - 1))).astype(np.float32)
    ref = oracle.sdf_to_alpha_fwd(msd, z, msl, s, clip)
    close(host(F.sdf_to_alpha_fwd(dev(msd), dev(z), dev(msl), s, clip=clip)), ref, rtol=1e-6, atol=5e-7)
    gout = rng.random((R, P - 1)).astype(np.float32)
    r_sdf, r_slope, r_s = oracle.sdf_to_alpha_bwd(msd, z, msl, s, gout, clip)
    d_sdf, d_slope, d_s = F.sdf_to_alpha_bwd(dev(msd), dev(z), dev(msl), s, dev(gout), clip=clip)
    close(host(d_sdf), r_sdf, rtol=1e-4, atol=1e-5 * np.abs(r_sdf).max())
    close(host(d_slope), r_slope, rtol=1e-4, atol=1e-5 * np.abs(r_slope).max())
    assert abs(float(d_s) - r_s) <= 1e-3 * abs(r_s) + 1e-5    # fp32 tree sum of 3.8e5 terms vs the oracle's double


def test_compositor_last_transmittance_is_differentiable(F):
    """trans_shift[:, -1] (what a background model is blended with, full_model.py:278-330) carries gradient to sigma / alpha:
    against plain torch autograd on alpha_to_weights (ray_helper.py:596-620), sigma and alpha= inputs, both add_inf_z."""
    from arcnerf_amd.render.ray_helper import ray_marching
    rng = np.random.default_rng(31)
    R, P = 257, 70
    z = dev(np.sort(rng.random((R, P)).astype(np.float32) * 4 + 0.5, axis=-1))
    rad = dev(rng.random((R, P, 3)).astype(np.float32))
    gt, gc = dev(rng.normal(size=R).astype(np.float32)), dev(rng.normal(size=(R, 3)).astype(np.float32))

    def torch_ref(sigma, alpha, add_inf_z):
        if alpha is None:
            deltas = z[:, 1:] - z[:, :-1]
            deltas = torch.where(deltas.abs() < 1e-5, torch.zeros_like(deltas), deltas)   # ray_helper.py:531
            if add_inf_z:
                deltas = torch.cat([deltas, torch.full((R, 1), 1e10, device=z.device)], -1)
                sg, rd = sigma, rad
            else:
                sg, rd = sigma[:, :-1], rad[:, :-1]
            alpha, rd_ = 1 - torch.exp(-torch.relu(sg) * deltas), rd
        else:
            rd_ = rad
        ts = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
        rgb = ((alpha * ts).unsqueeze(-1) * rd_).sum(-2)
        return (rgb * gc).sum() + (ts[:, -1] * gt).sum()

    for mode in ('sigma', 'sigma_inf', 'alpha'):
        leaf = dev((rng.random((R, P)) * (3.0 if mode != 'alpha' else 0.6)).astype(np.float32)).requires_grad_(True)
        ref_leaf = leaf.detach().clone().requires_grad_(True)
        if mode == 'alpha':
            out = ray_marching(None, rad, z, alpha=leaf)
            torch_ref(None, ref_leaf, False).backward()
        else:
            out = ray_marching(leaf, rad, z, add_inf_z=(mode == 'sigma_inf'))
            torch_ref(ref_leaf, None, mode == 'sigma_inf').backward()
        assert out['trans_shift'].requires_grad
        ((out['rgb'] * gc).sum() + (out['trans_shift'][:, -1] * gt).sum()).backward()
        close(host(leaf.grad), host(ref_leaf.grad), rtol=2e-4, atol=2e-5 * float(ref_leaf.grad.abs().max()))


# ---- hash grid, second order (NeuS on a hash grid: a loss on d encode / d xyz) -------------------------------------------
@pytest.mark.parametrize('tag', ['ngp', 'tiny', 'f4'])
def test_hashgrid_second_order_vs_reference_golden_and_oracle(F, oracle, tag):
    """G17 (reference torch backend under double autograd): the kernel behind arcn_hashgrid_bwd_bwd, and the autograd graph of
    HashGridEmbedder (table node + xyz node) differentiated twice."""
    from arcnerf_amd import _native as N
    from arcnerf_amd.ops.autograd import hashgrid_encode
    g = load_golden('g17_hashgrid_second_order')
    L, Fe, T, base, mx_res = [int(v) for v in g[tag + '_cfg']]
    res, offs = oracle.hashgrid_levels(L, T, base, mx_res)
    table = make_table(int(offs[-1]), Fe, seed=17, scale=0.5)
    desc = N.make_hashgrid_desc([int(r) for r in res], [int(o) for o in offs], Fe, [-1.0] * 3, [1.0] * 3)
    xyz, gy, gdx = g[tag + '_xyz'], g[tag + '_gy'], g[tag + '_gdx']
    ddout, dtable, d2x = F.hashgrid_bwd_bwd(dev(xyz), dev(gdx), dev(table), dev(gy), desc, want_d2xyz=True)   # binned table scatter (F <= 2)
    _, dtable_atomic, _ = F.hashgrid_bwd_bwd(dev(xyz), dev(gdx), dev(table), dev(gy), desc, want_ddout=False, workspace=None)
    close(host(dtable), host(dtable_atomic), rtol=1e-4, atol=1e-6 * float(dtable_atomic.abs().max()))
    o_ddout, o_dtable, o_d2x = oracle.hashgrid_bwd_bwd(xyz, gdx, table, gy, res, offs, np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32))
    for got, orc_v, key in ((ddout, o_ddout, '_d_gy'), (d2x, o_d2x, '_d_x')):
        scale = np.abs(g[tag + key]).max()
        close(host(got), orc_v, rtol=1e-4, atol=1e-5 * scale)
        close(host(got), g[tag + key], rtol=1e-4, atol=2e-4 * scale)
    rows = g[tag + '_d_table_rows']
    scale = np.abs(g[tag + '_d_table_vals']).max()
    close(host(dtable), o_dtable, rtol=1e-4, atol=1e-5 * scale)
    close(host(dtable)[rows], g[tag + '_d_table_vals'], rtol=1e-4, atol=1e-4 * scale)
    # the same numbers through autograd: dx with create_graph, then a second differentiation
    x = dev(xyz).requires_grad_(True)
    tab = dev(table).requires_grad_(True)
    gy_t = dev(gy).requires_grad_(True)
    y = hashgrid_encode(x, tab, desc)
    dx, = torch.autograd.grad((y * gy_t).sum(), x, create_graph=True)
    close(host(dx), g[tag + '_dx'], rtol=1e-4, atol=1e-4 * np.abs(g[tag + '_dx']).max())
    a_tab, a_gy, a_x = torch.autograd.grad((dx * dev(gdx)).sum(), [tab, gy_t, x])
    close(host(a_gy), g[tag + '_d_gy'], rtol=1e-4, atol=2e-4 * np.abs(g[tag + '_d_gy']).max())
    close(host(a_x), g[tag + '_d_x'], rtol=1e-4, atol=2e-4 * np.abs(g[tag + '_d_x']).max())
    close(host(a_tab)[rows], g[tag + '_d_table_vals'], rtol=1e-4, atol=1e-4 * scale)
    # first order is unchanged by the node split: table gradient on one node, xyz gradient on the other
    y2 = hashgrid_encode(x, tab, desc)
    b_tab, b_x = torch.autograd.grad((y2 * dev(gy)).sum(), [tab, x])
    ref_tab, ref_dx = oracle.hashgrid_bwd(xyz, table, gy, res, offs, np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32), want_dxyz=True)
    close(host(b_x), ref_dx, rtol=1e-4, atol=1e-5 * np.abs(ref_dx).max())
    close(host(b_tab), ref_tab, rtol=1e-4, atol=1e-5 * np.abs(ref_tab).max())
    # a density model never builds the xyz node
    y3 = hashgrid_encode(dev(xyz), tab, desc)
    assert type(y3.grad_fn).__name__ == 'HashGridFnBackward'


# ---- ray generation -----------------------------------------------------------------------------------------------------
def test_get_rays_vs_reference_golden_and_oracle(F, oracle):
    """G20 (reference get_rays): both flattening orders with the mip-nerf radius, pixel subsets, centre pixel, raw directions, NDC"""
    from arcnerf_amd.render.ray_helper import get_rays
    g = load_golden('g20_get_rays')
    W, H, K, c2w, idx = int(g['W']), int(g['H']), g['K'], g['c2w'], g['index']
    cases = {'wh': dict(wh_order=True), 'hw': dict(wh_order=False), 'center': dict(wh_order=True, center_pixel=True),
             'raw': dict(wh_order=False, normalize_rays_d=False), 'ndc': dict(wh_order=True, ndc=True, ndc_near=1.0),
             'idx': dict(index=idx), 'idx_center_ndc': dict(index=idx, center_pixel=True, ndc=True, ndc_near=0.5)}
    for tag, kw in cases.items():
        kw_t = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        o, d, flat, r = get_rays(W, H, dev(K), dev(c2w), **kw_t)
        ro, rd, rr = oracle.get_rays(W, H, K, c2w, **kw)
        close(host(o), g[tag + '_o'], rtol=1e-5, atol=1e-5)
        close(host(d), g[tag + '_d'], rtol=1e-5, atol=1e-5)
        close(host(o), ro, rtol=1e-6, atol=1e-6)
        close(host(d), rd, rtol=1e-6, atol=1e-6)
        if tag + '_r' in g.files:
            assert flat is None
            close(host(r), g[tag + '_r'], rtol=1e-4, atol=1e-6)
            close(host(r), rr, rtol=1e-5, atol=1e-7)
        else:
            assert r is None and flat == g[tag + '_flat'].tolist()
    # Lego-sized image: the full 800x800 grid in one launch, unit directions, origin = camera centre
    o, d, _, r = get_rays(800, 800, dev(K) * 60, dev(c2w))
    assert o.shape == (640000, 3) and torch.allclose(d.norm(dim=-1), torch.ones(640000, device='cuda'), atol=1e-5)
    assert torch.equal(o, dev(c2w)[:3, 3].expand_as(o)) and r.shape == (640000, 1) and float(r.min()) > 0
    np.random.seed(0)
    o, d, flat, r = get_rays(W, H, dev(K), dev(c2w), n_rays=20)
    assert o.shape == (20, 3) and len(set(flat)) == 20 and r is None
    with pytest.raises(AssertionError):
        get_rays(W, H, dev(K), dev(c2w), index=dev(idx), n_rays=5)


def test_sample_pdf_one_kernel_vs_oracle_and_reference(F, oracle):
    """arcn_sample_pdf (weights -> cdf with torch's CPU cumsum semantics -> inverse CDF -> sort): cdf and samples bit-identical to the
    oracle's restatement for the lattice and for per-ray uniforms; against golden G2 (the reference's own cdf / samples) the cdf agrees
    to the normaliser's last ulp and the samples wherever no decision sits on a tie"""
    g = load_golden('g2_resampling')
    bins, w = g['bins'], g['weights']
    for u in (g['u_det'][:1], g['u_det'], g['u']):
        s, cdf = F.sample_pdf(dev(bins), dev(w), dev(np.ascontiguousarray(u)), want_cdf=True)
        ref_cdf = oracle.weights_to_cdf(w)
        assert np.array_equal(host(cdf).view(np.uint32), ref_cdf.view(np.uint32))
        uu = np.broadcast_to(u, (bins.shape[0], u.shape[1]))
        ref_s = oracle.sample_cdf(bins, ref_cdf, np.ascontiguousarray(uu))[0]
        assert np.array_equal(host(s).view(np.uint32), ref_s.view(np.uint32))
        close(host(cdf), g['cdf'], rtol=3e-7, atol=1.5e-7)
    bad = np.abs(host(F.sample_pdf(dev(bins), dev(w), dev(np.ascontiguousarray(g['u_det'][:1])))) - g['samples_det']) > 1e-4
    assert bad.mean() < 2e-3
    # ragged sizes: 2 bins, 1 sample; many samples
    rng = np.random.default_rng(0)
    for n_pts, n_s in ((2, 1), (3, 7), (193, 128), (1025, 64)):
        b = np.sort(rng.random((5, n_pts)).astype(np.float32), axis=1)
        ww = (rng.random((5, n_pts - 1)) ** 8).astype(np.float32)
        u = rng.random((5, n_s)).astype(np.float32)
        s, cdf = F.sample_pdf(dev(b), dev(ww), dev(u), want_cdf=True)
        ref_cdf = oracle.weights_to_cdf(ww)
        assert np.array_equal(host(cdf).view(np.uint32), ref_cdf.view(np.uint32))
        assert np.array_equal(host(s).view(np.uint32), oracle.sample_cdf(b, ref_cdf, u)[0].view(np.uint32))


def test_fused_adam_optimizer_matches_torch_adam(F):
    """arcnerf_amd.optim.FusedAdam (torch.optim front end of arcn_adam_ema_step) against torch.optim.Adam: several steps, weight
    decay, ragged sizes, changing gradients; state_dict layout; fused gradient clear"""
    from arcnerf_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 2), (33, 64), (7,)]
    pa = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-3, zero_grad_on_step=True)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-3)
    for it in range(12):
        for a, b_ in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (0.1 if it % 3 else 10.0)
            a.grad = gr.clone()
            b_.grad = gr.clone()
        oa.step()
        ob.step()
        for a in pa:
            assert float(a.grad.abs().max()) == 0.0   # cleared in the same pass
    for a, b_ in zip(pa, pb):
        close(host(a), host(b_), rtol=2e-6, atol=2e-7)
    sd = oa.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'} and sd['state'][0]['step'] == 12
    close(host(sd['state'][1]['exp_avg_sq']), host(ob.state_dict()['state'][1]['exp_avg_sq']), rtol=2e-6, atol=1e-12)
    with pytest.raises(RuntimeError):
        cpu_p = torch.zeros(4, requires_grad=True)
        cpu_p.grad = torch.ones(4)
        FusedAdam([cpu_p]).step()
    # the EMA variants: shadow in state['ema'] vs in the parameter itself, same bits
    qa = [p.detach().clone().requires_grad_(True) for p in pb]
    qb = [p.detach().clone().requires_grad_(True) for p in pb]
    ea = FusedAdam(qa, lr=1e-1, eps=1e-15, ema_decay=0.95)
    eb = FusedAdam(qb, lr=1e-1, eps=1e-15, ema_decay=0.95, ema_in_param=True)
    for it in range(4):
        for a, b_ in zip(qa, qb):
            a.grad = torch.randn(a.shape, generator=g).cuda()
            b_.grad = a.grad.clone()
        ea.step()
        eb.step()
    for a, b_ in zip(qa, qb):
        assert torch.equal(a, b_) and torch.equal(ea.state[a]['ema'], a)
    assert 'ema' not in eb.state[qb[0]]


def test_fused_adam_flat_buffer_equals_per_parameter_steps(F):
    """FusedAdam.flatten(): parameters, gradients (views that autograd accumulates into) and state in ONE buffer per group, one launch
    per step, `grad_scale` = DDP's 1 / world on a SUM all-reduced buffer; same bits as the per-parameter launches, torch.optim.Adam's
    state_dict layout, load_state_dict re-flattens, a re-assigned .grad is refused"""
    from arcnerf_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(257, 63), (33, 64), (7,), (1,), (64, 3)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    xs = [torch.randn(16, s[-1] if len(s) > 1 else s[0], generator=g).cuda() for s in shapes]

    def loss(ps, k):
        return sum(((x @ p.t() if p.dim() > 1 else x * p) * (k + 1.0)).pow(2).mean() for x, p in zip(xs, ps))
    oa = FusedAdam(pa, lr=1e-2, eps=1e-15, weight_decay=1e-3, ema_decay=0.95).flatten()
    ob = FusedAdam(pb, lr=1e-2, eps=1e-15, weight_decay=1e-3, ema_decay=0.95)
    oa.grad_scale = ob.grad_scale = 0.5
    fg = oa.flat_grads()
    assert fg.numel() % 4 == 0 and all(p.grad.data_ptr() % 16 == 0 for p in pa)
    for it in range(5):
        oa.zero_grad()
        ob.zero_grad(set_to_none=False)
        loss(pa, it).backward()
        loss(pb, it).backward()
        assert fg.data_ptr() == oa.flat_grads().data_ptr() and float(fg.abs().sum()) > 0     # autograd wrote into the flat buffer
        for a, b_ in zip(pa, pb):
            assert torch.equal(a.grad, b_.grad)
        oa.step()
        ob.step()
        for a, b_ in zip(pa, pb):
            assert torch.equal(a, b_)
    sd = oa.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq', 'ema'} and sd['state'][0]['step'] == 5
    assert torch.equal(sd['state'][2]['exp_avg_sq'], ob.state_dict()['state'][2]['exp_avg_sq'])
    # a checkpoint goes back in: the loaded tensors are folded into the flat buffers again
    oa.load_state_dict(ob.state_dict())
    oa.zero_grad()
    ob.zero_grad(set_to_none=False)
    loss(pa, 9).backward()
    loss(pb, 9).backward()
    oa.step()
    ob.step()
    for a, b_ in zip(pa, pb):
        assert torch.equal(a, b_)
    pa[1].grad = torch.zeros_like(pa[1])
    with pytest.raises(RuntimeError):
        oa.step()


def test_direct_gradients_only_when_the_engine_accumulates(F):
    """Parameters re-homed by FusedAdam.flatten() let hand-written backward nodes add dW straight into the flat gradient buffer - but only
    in an engine run that WOULD accumulate into .grad: `torch.autograd.grad(y, x)` (BaseGeoNet.forward_with_grad, the normals
    of a geometry net evaluated between steps) leaves the buffer untouched, `torch.autograd.grad(y, params)` returns the gradient, and
    `backward()` puts the same gradient into the buffer; a non-default flatten(direct_grads=False) survives load_state_dict."""
    from arcnerf_amd.models.base_modules.geo_rad_model.tcnn_fusedmlp_module import FusedLayers
    from arcnerf_amd.optim import FusedAdam
    torch.manual_seed(3)
    layers = FusedLayers([32, 64, 16], 'ReLU', 'None').cuda()
    opt = FusedAdam(layers.parameters(), lr=1e-2).flatten()
    fg = opt.flat_grads()
    x = torch.randn(1000, 32, device='cuda', requires_grad=True)
    gx, = torch.autograd.grad(layers(x).sum(), x)
    assert float(gx.abs().max()) > 0 and float(fg.abs().max()) == 0.0
    gp, = torch.autograd.grad(layers(x).sum(), layers.params)
    assert float(gp.abs().max()) > 0 and float(fg.abs().max()) == 0.0
    layers(x).sum().backward()
    assert layers.params.grad.data_ptr() == fg.data_ptr()
    assert float((layers.params.grad - gp).abs().max()) <= 1e-5 * float(gp.abs().max())
    opt2 = FusedAdam(FusedLayers([32, 64, 16], 'ReLU', 'None').cuda().parameters(), lr=1e-2).flatten(direct_grads=False)
    opt2.load_state_dict(opt2.state_dict())
    assert all(p._arcn_direct_grad is False for g in opt2.param_groups for p in g['params'])


def test_first_order_only_nodes_refuse_a_second_differentiation(F):
    """the fused MLP has a hand-written first-order backward: asking autograd to differentiate THROUGH that backward (an sdf net on
    the fused kernels with normals by create_graph) must raise, not return gradients that silently ignore the path"""
    from arcnerf_amd import _native as N
    from arcnerf_amd.ops.autograd import FusedMlpFn
    desc = N.make_mlp_desc([8, 16, 4], 'relu', None, False)
    w = torch.randn(8 * 16 + 16 * 4, device='cuda', requires_grad=True)
    x = torch.randn(64, 8, device='cuda', requires_grad=True)
    y = FusedMlpFn.apply(x, w, None, desc)
    dx, = torch.autograd.grad(y.sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):
        dx.pow(2).sum().backward()


def test_adam_ema_shadow_in_parameter_is_bit_identical(F):
    """ema.py:41-42 stores the new average in the parameter AND the shadow, so the two are equal after every step: passing the
    parameter buffer as `ema` (no second copy) must give the same bits as keeping the shadow - from the first step on, whatever the
    shadow held before it (the step-1 debias factor of the old average is 0)."""
    gpu = torch.device('cuda:0')
    n = 100003
    g = torch.Generator(device='cuda').manual_seed(11)
    p0 = torch.randn(n, device=gpu, generator=g)
    state = []
    for alias in (False, True):
        p, m, v = p0.clone(), torch.zeros(n, device=gpu), torch.zeros(n, device=gpu)
        shadow = p if alias else torch.full((n,), 7.0, device=gpu)
        gg = torch.Generator(device='cuda').manual_seed(5)
        for step in range(1, 6):
            grad = torch.randn(n, device=gpu, generator=gg)
            F.adam_ema_step(p, grad, m, v, shadow, step, lr=1e-1, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6, ema_decay=0.95,
                            zero_grad=True)
            assert float(grad.abs().max()) == 0.0
            if not alias:
                assert torch.equal(shadow, p)
        state.append((p, m, v))
    for a, b in zip(*state):
        assert torch.equal(a, b)


@pytest.mark.parametrize('add_inf_z', [True, False])
def test_ray_marching_weights_gradient_vs_torch(add_inf_z):
    """A loss on the per-sample `weights` output (e.g. a normal map = sum w * n) differentiates through the fused compositor like
    the reference's torch ops (ray_helper.py:476-620); together with rgb / depth / mask gradients; no host read in backward."""
    from arcnerf_amd.render.ray_helper import ray_marching
    g = torch.Generator().manual_seed(7)
    R, P = 33, 41
    sigma = (torch.rand(R, P, generator=g) * 6.0 - 1.0).cuda().requires_grad_(True)
    rad = torch.rand(R, P, 3, generator=g).cuda().requires_grad_(True)
    z = torch.sort(torch.rand(R, P, generator=g) * 4.0 + 1.0, -1)[0].cuda()
    vec = torch.randn(R, P if add_inf_z else P - 1, 3, generator=g).cuda()
    bkg = torch.rand(R, 3, generator=g).cuda()

    def torch_ref(sg, rd):
        deltas = z[:, 1:] - z[:, :-1]
        deltas = torch.where(deltas.abs() < 1e-5, torch.zeros_like(deltas), deltas)
        if add_inf_z:
            deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
            s_, r_, z_ = sg, rd, z
        else:
            s_, r_, z_ = sg[:, :-1], rd[:, :-1], z[:, :-1]
        alpha = 1 - torch.exp(-torch.relu(s_) * deltas)
        trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * trans
        rgb = (w[..., None] * r_).sum(-2) + trans[:, -1:] * bkg
        return rgb, (w * z_).sum(-1), w.sum(-1), w

    def loss_of(rgb, depth, mask, w):
        normal = (w[..., None] * vec).sum(1)
        return (rgb ** 2).sum() + 0.3 * depth.sum() + (mask ** 2).sum() + (normal ** 2).sum()

    out = ray_marching(sigma, rad, z, add_inf_z=add_inf_z, bkg_color=bkg)
    loss_of(out['rgb'], out['depth'], out['mask'], out['weights']).backward()
    gs, gr = sigma.grad.clone(), rad.grad.clone()
    sigma.grad = rad.grad = None
    rgb, depth, mask, w = torch_ref(sigma, rad)
    assert (out['weights'] - w).abs().max() < 1e-5
    loss_of(rgb, depth, mask, w).backward()
    assert (gs - sigma.grad).abs().max() <= 1e-4 * sigma.grad.abs().max() + 1e-6
    assert (gr - rad.grad).abs().max() <= 1e-4 * rad.grad.abs().max() + 1e-6
    # weights only (the other outputs undifferentiated arrive as None)
    sigma.grad = None
    (ray_marching(sigma, rad, z, add_inf_z=add_inf_z, bkg_color=bkg)['weights'] ** 2).sum().backward()
    g1 = sigma.grad.clone()
    sigma.grad = None
    (torch_ref(sigma, rad)[3] ** 2).sum().backward()
    assert (g1 - sigma.grad).abs().max() <= 1e-4 * sigma.grad.abs().max() + 1e-6


GEMM_SHAPES = [(5000, 63, 256), (1001, 319, 256), (777, 256, 257), (4096, 283, 128), (3000, 128, 3), (2500, 32, 64), (2500, 64, 17),
               (130, 1, 128), (64, 128, 1), (9, 39, 217)]


@pytest.mark.parametrize('S,K,Nn', GEMM_SHAPES)
def test_gemm_products_vs_float64(S, K, Nn):
    """arcn_gemm_nt / nn / tn (csrc/gemm.hip: the dense layers of GeoNet / RadianceNet, linear_network_module.py:174-197,318-335) against
    float64 matmuls: exact f32 MFMA accumulation, so only summation-order noise."""
    from arcnerf_amd.ops import functional as Fn
    g = torch.Generator().manual_seed(S + K + Nn)
    x = torch.randn(S, K, generator=g).cuda()
    w = (torch.randn(Nn, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(Nn, generator=g).cuda()
    dy = torch.randn(S, Nn, generator=g).cuda()
    y = Fn.gemm_nt(x, w, b)
    ref = (x.double() @ w.double().t() + b.double())
    assert (y.double() - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item())
    yr = Fn.gemm_nt(x, w, None, act='relu')
    assert (yr.double() - torch.relu(x.double() @ w.double().t())).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item())
    # softplus(beta = 100) epilogue (the graph-free sdf evaluations of NeuS): hardware exp2 / log2 with a compensated log1p
    ysp = Fn.gemm_nt(x * 0.05, w, b * 0.05, act='softplus', beta=100.0)
    refsp = torch.nn.functional.softplus((x.double() * 0.05) @ w.double().t() + b.double() * 0.05, beta=100.0)
    assert (ysp.double() - refsp).abs().max() <= 2e-6 * max(1.0, refsp.abs().max().item())
    # relative too (tiny values far below zero): beta x the product's own rounding (2e-6 of its scale) is the floor, e^(beta z) amplifies it
    assert ((ysp.double() - refsp).abs() <= 3e-4 * refsp.abs() + 1e-9).all()
    dx = Fn.gemm_nn(dy, w)
    refx = dy.double() @ w.double()
    assert (dx.double() - refx).abs().max() <= 2e-6 * max(1.0, refx.abs().max().item())
    dw = Fn.gemm_tn(dy, x)
    refw = dy.double().t() @ x.double()
    assert (dw.double() - refw).abs().max() <= 2e-6 * max(1.0, refw.abs().max().item()) * max(1.0, (S / 1000.0) ** 0.5)
    assert torch.equal(dw, Fn.gemm_tn(dy, x))      # fixed reduction order: bit-reproducible
    # masked operand (a ReLU layer's backward) + the bias gradient from the same pass
    m = torch.randn(S, Nn, generator=g).cuda()
    dyd = dy.double() * (m > 0)
    if K % 4 == 0 and Nn % 4 == 0:
        dwm, db = Fn.gemm_tn(dy, x, mask=m, want_colsum=True)
        refw = dyd.t() @ x.double()
        assert (dwm.double() - refw).abs().max() <= 2e-6 * max(1.0, refw.abs().max().item()) * max(1.0, (S / 1000.0) ** 0.5)
        refb = dyd.sum(0)
        assert (db.double() - refb).abs().max() <= 2e-6 * max(1.0, refb.abs().max().item()) * max(1.0, (S / 1000.0) ** 0.5)
        assert torch.equal(dwm, Fn.gemm_tn(dy, x, mask=m))
        dxm = Fn.gemm_nn(dy, w, mask=m)
        refx = dyd @ w.double()
        assert (dxm.double() - refx).abs().max() <= 2e-6 * max(1.0, refx.abs().max().item())


@pytest.mark.parametrize('S,K,Nn', [(3001, 256, 256), (1111, 320, 256), (2000, 284, 128), (777, 128, 384), (130, 68, 132), (7, 108, 192), (1, 204, 224),
                                    (131077, 336, 272)])
def test_relu_bit_masks_equal_float_masks(S, K, Nn):
    """arcn_gemm_nt_split(relu_bits): word [s / 8][f / 4] bit 4 (s % 8) + f % 4 is (y[s][f] > 0); the masked gradient products give
    bit-identical results whether they read those words (mask_bits) or y itself as the mask (linear.py:11-35's ReLU backward)."""
    from arcnerf_amd.ops import functional as Fn
    g = torch.Generator().manual_seed(S + K)
    x = torch.randn(S, K, generator=g).cuda()
    w = (torch.randn(Nn, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(Nn, generator=g).cuda()
    dy = torch.randn(S, Nn, generator=g).cuda()
    assert Fn.relu_bits_supported(x, K, Nn)
    y, bits = Fn.gemm_nt(x, w, b, act='relu', want_bits=True)
    # (same product; a <= 16-wide remainder of the outputs runs on the exact-f32 kernel only when no bits are asked for)
    assert (y - Fn.gemm_nt(x, w, b, act='relu')).abs().max() <= 2e-6 * y.abs().max()
    So = (S + 7) // 8
    pos = torch.zeros(So * 8, Nn, dtype=torch.int64, device='cuda')
    pos[:S] = (y > 0)
    blocks = pos.view(So, 8, Nn // 4, 4).permute(0, 2, 1, 3).reshape(So, Nn // 4, 32)      # word [s / 8][f / 4], bit 4 (s % 8) + f % 4
    words = (blocks << torch.arange(32, device='cuda')).sum(-1)
    assert torch.equal(bits.to(torch.int64) & 0xffffffff, words)
    assert torch.equal(Fn.gemm_nn(dy, w, mask_bits=bits), Fn.gemm_nn(dy, w, mask=y))
    dw_b, db_b = Fn.gemm_tn(dy, x, mask_bits=bits, want_colsum=True)
    dw_f, db_f = Fn.gemm_tn(dy, x, mask=y, want_colsum=True)
    assert torch.equal(dw_b, dw_f) and torch.equal(db_b, db_f)
    assert torch.equal(Fn.gemm_tn(dy, x, mask_bits=bits), Fn.gemm_tn(dy, x, mask=y))


@pytest.mark.parametrize('n', [(4096, 256), (1001, 217), (7,)])
def test_softplus_twice_differentiable_vs_torch(n):
    """ops.autograd.softplus (SoftplusFn / SoftplusGradFn on arcn_softplus_grad / arcn_softplus_grad2) against
    torch.nn.functional.softplus(beta=100) under the NeuS pattern: value, d / d input with create_graph, then the gradients of a loss on
    both through the first derivative (base_network.py:30-44 over nn.Softplus(beta=100), activation.py)."""
    from arcnerf_amd.ops.autograd import softplus
    g = torch.Generator().manual_seed(len(n))
    z0 = (torch.randn(*n, generator=g) * 0.08).cuda()
    z0.view(-1)[:3] = torch.tensor([0.5, -0.5, 0.2])           # beyond / far below / at the threshold beta z = 20
    w0 = torch.randn(*n, generator=g).cuda()
    res = {}
    for name, fn in (('hip', lambda t: softplus(t, 100.0)), ('torch', lambda t: torch.nn.functional.softplus(t, beta=100.0))):
        z = z0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        y = fn(z * w)
        dz, = torch.autograd.grad(y, z, torch.ones_like(y) * 0.7, create_graph=True)
        loss = (y ** 2).sum() + (dz ** 2).sum() * 0.3 + (dz * w).sum()
        gz, gw = torch.autograd.grad(loss, (z, w))
        res[name] = (y.detach(), dz.detach(), gz, gw)
    for a, b, tol in zip(res['hip'], res['torch'], (1e-6, 1e-5, 1e-4, 1e-4)):
        assert (a - b).abs().max() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max()


@pytest.mark.parametrize('n,C,W', [(5000, 3, 128), (257, 3, 128), (1, 1, 7), (70001, 2, 64)])
def test_tone_mappers_vs_torch(n, C, W):
    """arcn_tonemap_fwd / bwd (ops.autograd.ToneMapFn) against the reference's per-channel DenseLayer(1, W) + ReLU, DenseLayer(W, 1) +
    sigmoid stacks (hdrnerf_model.py:44-75) under torch autograd: values, input gradient, parameter gradients."""
    from arcnerf_amd.ops.autograd import ToneMapFn
    g = torch.Generator().manual_seed(n + W)
    x0 = (torch.randn(n, C, generator=g) * 2.0).cuda()
    p0 = torch.randn(C, 3 * W + 1, generator=g).cuda() * 0.5
    gy = torch.randn(n, C, generator=g).cuda()
    x, p = x0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    y = ToneMapFn.apply(x, p)
    y.backward(gy)
    xr, pr = x0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    cols = []
    for c in range(C):
        w1, b1, w2, b2 = pr[c, :W], pr[c, W:2 * W], pr[c, 2 * W:3 * W], pr[c, 3 * W]
        h = torch.relu(xr[:, c:c + 1] * w1[None] + b1[None])
        cols.append(torch.sigmoid(h @ w2 + b2))
    yr = torch.stack(cols, -1)
    yr.backward(gy)
    assert (y - yr).abs().max() <= 2e-6
    assert (x.grad - xr.grad).abs().max() <= 2e-5 * max(1.0, xr.grad.abs().max().item())
    assert (p.grad - pr.grad).abs().max() <= 1e-4 * max(1.0, pr.grad.abs().max().item()), (p.grad - pr.grad).abs().max()
    x2, p2 = x0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    ToneMapFn.apply(x2, p2).backward(gy)
    assert torch.equal(p2.grad, p.grad)        # fixed summation order


def test_sdf_net_with_explicit_jacobian_vs_torch_double_backward():
    """ops.autograd.SdfMlpJacFn (out = W2 softplus(W1 f) and J = d out[:, 0] / d f as explicit outputs of a first-order node) against the
    reference's construction - torch.autograd.grad(..., create_graph=True) through the layer stack (base_network.py:30-44) - for a loss on
    both: values and the gradients with respect to the features and both weight matrices."""
    from arcnerf_amd.ops.autograd import SdfMlpJacFn
    g = torch.Generator().manual_seed(3)
    S, K, H, O, beta = 5001, 32, 64, 20, 100.0
    f0 = (torch.randn(S, K, generator=g) * 0.3).cuda()
    w10 = (torch.randn(H, K, generator=g) / K ** 0.5 * 0.3).cuda()
    w20 = (torch.randn(O, H, generator=g) / H ** 0.5).cuda()
    go, gj = torch.randn(S, O, generator=g).cuda(), torch.randn(S, K, generator=g).cuda()
    res = {}
    for name in ('hip', 'torch'):
        f, w1, w2 = (t.clone().requires_grad_(True) for t in (f0, w10, w20))
        if name == 'hip':
            out, jac = SdfMlpJacFn.apply(f, w1, w2, beta)
        else:
            out = torch.nn.functional.softplus(f @ w1.t(), beta=beta) @ w2.t()
            jac, = torch.autograd.grad(out[:, 0].sum(), f, create_graph=True)
        loss = (out * go).sum() + (jac * gj).sum()
        res[name] = (out.detach(), jac.detach()) + torch.autograd.grad(loss, (f, w1, w2))
    for a, b in zip(res['hip'], res['torch']):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max(), ((a - b).abs().max() / b.abs().max()).item()


def test_linear_layers_double_backward_vs_torch():
    """a 3-layer softplus(100) net with a skip concat on ops.autograd.linear: outputs, d out / d x (create_graph), and the gradients of a
    loss on BOTH (the NeuS pattern: rgb loss + Eikonal on the normals) against the same net on torch.nn.functional.linear."""
    from arcnerf_amd.ops.autograd import linear
    g = torch.Generator().manual_seed(5)
    S = 3000
    x0 = torch.randn(S, 39, generator=g).cuda()
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).cuda().requires_grad_(True) for o, i in ((256, 39), (217, 256), (65, 256))]
    bs = [(torch.randn(o, generator=g) * 0.1).cuda().requires_grad_(True) for o in (256, 217, 65)]

    def net(x, lin):
        h = torch.nn.functional.softplus(lin(x, Ws[0], bs[0]), beta=100)
        h = torch.nn.functional.softplus(lin(h, Ws[1], bs[1]), beta=100)
        h = torch.cat([h, x], -1) / 2 ** 0.5
        return lin(h, Ws[2], bs[2])

    res = {}
    for name, lin in (('hip', linear), ('torch', torch.nn.functional.linear)):
        x = x0.clone().requires_grad_(True)
        out = net(x, lin)
        sdf = out[:, :1]
        nrm = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True)[0]
        loss = (out[:, 1:] ** 2).mean() + 0.1 * ((nrm[:, :3].norm(dim=-1) - 1.0) ** 2).mean()
        for p in Ws + bs:
            p.grad = None
        loss.backward()
        res[name] = [out.detach(), nrm.detach(), x.grad.clone()] + [p.grad.clone() for p in Ws + bs]
    for a, b_ in zip(res['hip'], res['torch']):
        assert (a - b_).abs().max() <= 2e-4 * b_.abs().max() + 1e-7


def test_relu_layers_double_backward_vs_torch():
    """the fused DenseLayer (LinearReluFn: bias + ReLU in the epilogue, bit masks in the backward) under create_graph: an sdf / geometry
    net with ReLU hidden layers whose input gradient (the normal) enters the loss - the reference's layers are plain torch ops,
    differentiable to any order (base_network.py:30-44); wide layers (bit-mask path) and narrow ones (float-mask path)"""
    from arcnerf_amd.ops.autograd import linear, linear_relu
    g = torch.Generator().manual_seed(9)
    S = 2048
    x0 = torch.randn(S, 40, generator=g).cuda()
    Ws = [(torch.randn(o, i, generator=g) / i ** 0.5).cuda().requires_grad_(True) for o, i in ((256, 40), (64, 256), (9, 64))]
    bs = [(torch.randn(o, generator=g) * 0.1).cuda().requires_grad_(True) for o in (256, 64, 9)]

    def net(x, hip):
        lr = linear_relu if hip else (lambda a, w, b: torch.relu(torch.nn.functional.linear(a, w, b)))
        ln = linear if hip else torch.nn.functional.linear
        return ln(lr(lr(x, Ws[0], bs[0]), Ws[1], bs[1]), Ws[2], bs[2])

    res = {}
    for name in ('hip', 'torch'):
        x = x0.clone().requires_grad_(True)
        out = net(x, name == 'hip')
        sdf = out[:, :1]
        nrm = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True)[0]
        loss = (out[:, 1:] ** 2).mean() + 0.1 * ((nrm[:, :3].norm(dim=-1) - 1.0) ** 2).mean()
        for p in Ws + bs:
            p.grad = None
        loss.backward()
        res[name] = [out.detach(), nrm.detach(), x.grad.clone()] + [p.grad.clone() for p in Ws + bs]
    for a, b_ in zip(res['hip'], res['torch']):
        assert (a - b_).abs().max() <= 2e-4 * b_.abs().max() + 1e-7
    # a gradient that arrives as a view at an odd offset (not 16-byte aligned) must not trip the bit-mask products
    x = x0.clone().requires_grad_(True)
    y = linear_relu(x, Ws[0], bs[0])
    up = torch.randn(S * 256 + 1, generator=g).cuda()[1:].view(S, 256)
    assert up.data_ptr() % 16 != 0
    gx, = torch.autograd.grad(y, x, up)
    ref, = torch.autograd.grad(torch.relu(torch.nn.functional.linear(x, Ws[0], bs[0])), x, up)
    assert (gx - ref).abs().max() <= 2e-4 * ref.abs().max()


@pytest.mark.parametrize('S,K,Nn,bias', [(3000, 63, 256, True), (2049, 319, 256, True), (1500, 283, 128, True), (1000, 32, 64, False), (777, 256, 257, True)])
def test_fused_linear_relu_layer_vs_torch(S, K, Nn, bias):
    """LinearReluFn (bias + ReLU in the epilogue of arcn_gemm_nt, the ReLU mask folded into arcn_gemm_nn / arcn_gemm_tn, bias gradient on
    the reduction kernel) against relu(F.linear) under torch autograd: DenseLayer of linear.py:11-35."""
    from arcnerf_amd.ops.autograd import linear_relu
    g = torch.Generator().manual_seed(K * 7 + Nn)
    x0 = torch.randn(S, K, generator=g).cuda()
    w = (torch.randn(Nn, K, generator=g) / K ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Nn, generator=g) * 0.3).cuda().requires_grad_(True) if bias else None
    up = torch.randn(S, Nn, generator=g).cuda()
    x = x0.clone().requires_grad_(True)
    y = linear_relu(x, w, b)
    (y * up).sum().backward()
    ref_y = torch.relu(torch.nn.functional.linear(x0.double(), w.detach().double(), None if b is None else b.detach().double()))
    assert (y.double() - ref_y).abs().max() <= 2e-6 * ref_y.abs().max()
    # the gradients against float64 products with the layer's OWN activation mask (an entry whose pre-activation is within rounding
    # of zero may fall on either side of the ReLU in two different summation orders; its gradient then legitimately differs)
    dpre = (up * (y.detach() > 0)).double()
    for got, ref in ((x.grad, dpre @ w.detach().double()), (w.grad, dpre.t() @ x0.double())) + (((b.grad, dpre.sum(0)),) if bias else ()):
        assert got.shape == ref.shape
        assert (got.double() - ref).abs().max() <= 3e-6 * ref.abs().max() * max(1.0, (S / 1000.0) ** 0.5)


@pytest.mark.parametrize('side', [2.0, 1.5, 24.0])
def test_balanced_gather_is_bit_identical_for_any_launch_size(side):
    """arcn_hashgrid_fwd_xcd (cost-balanced XCD plan, reciprocal division with exact fallback, shared modulo, 16-byte pair loads) against
    the plain gather for 1 ... 2.6e5 points incl. the volume's corners and faces: every feature bit for bit, row- and level-major.
    (A launch with fewer workgroups than plan segments used to be rejected.)"""
    import ctypes as C
    from arcnerf_amd import _native as N
    from arcnerf_amd.ops import functional as Fn
    from arcnerf_amd.pipeline import hashgrid_level_table
    res, offs = hashgrid_level_table(16, 19, 16, 2048)
    lib, st = N.lib(), N.stream()
    h = side / 2
    desc = N.make_hashgrid_desc(res, offs, 2, [-h] * 3, [h] * 3)
    table = (torch.rand(offs[-1] * 2, device='cuda') - 0.5)
    for n in (1, 63, 300, 2048, 7257, 20000, 262144 + 77):
        g = torch.Generator().manual_seed(n)
        xyz = ((torch.rand(n, 3, generator=g) - 0.5) * side * 1.05).cuda()
        if n >= 5:
            xyz[:5] = torch.tensor([[-h, -h, -h], [h, h, h], [0, 0, 0], [h - 1e-7, 0, 0], [-h + 1e-7, 0.1, 0.2]], device='cuda')
        ref = Fn.hashgrid_fwd_plain(xyz, table, desc)
        assert torch.equal(Fn.hashgrid_fwd(xyz, table, desc), ref), n      # (the front end: level-major + transposing pass for the large launches)
        rm = torch.zeros(n, 32, device='cuda')
        lm = torch.zeros(16, n, 2, device='cuda')
        N.check(lib.arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(rm), 0, n, n, None, st))
        N.check(lib.arcn_hashgrid_fwd_xcd(N.ptr(xyz), N.ptr(table), C.addressof(desc), N.ptr(lm), 1, n, n, None, st))
        assert torch.equal(rm, ref), n
        assert torch.equal(lm.permute(1, 0, 2).reshape(n, 32), ref), n


@pytest.mark.gpu
@pytest.mark.parametrize('S,K,Nn,T,bias', [(1000, 256, 256, 64, True), (131, 64, 128, 4, False), (4099, 320, 256, 64, True), (77, 32, 32, 12, True)])
def test_layer_writing_into_the_skip_concatenation_vs_torch(S, K, Nn, T, bias):
    """LinearReluCatFn: cat([relu(F.linear(x, w, b)), tail]) with the product writing its columns of the concatenated buffer at that row
    stride and the gradient products reading theirs in place (GeoNet's skip, linear_network_module.py:174-197) - bit-identical with
    the layer followed by torch.cat, gradients of x, w, b and of the tail included."""
    from arcnerf_amd.ops.autograd import linear_relu, linear_relu_cat
    g = torch.Generator().manual_seed(S + Nn)
    x0 = torch.randn(S, K, generator=g).cuda()
    t0 = torch.randn(S, T, generator=g).cuda()
    w0 = (torch.randn(Nn, K, generator=g) / K ** 0.5).cuda()
    b0 = (torch.randn(Nn, generator=g) * 0.3).cuda() if bias else None
    up = torch.randn(S, Nn + T, generator=g).cuda()

    def run(fused):
        x, t, w = x0.clone().requires_grad_(True), t0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True) if bias else None
        out = linear_relu_cat(x, w, b, t) if fused else torch.cat([linear_relu(x, w, b), t], dim=-1)
        assert out is not None and out.shape == (S, Nn + T)
        (out * up).sum().backward()
        return [out.detach(), x.grad, w.grad, t.grad] + ([b.grad] if bias else [])
    for a, b_ in zip(run(True), run(False)):
        assert torch.equal(a, b_)


@pytest.mark.gpu
def test_weights_split_once_for_all_chunks_is_bit_identical():
    """chunk_processing opens ops.functional.split_weight_scope: a layer evaluated on several chunks takes ONE split of its weights
    (arcn_gemm_split_weights + ws_ready = 1) and one zero-padded copy of an odd-width weight; outputs and gradients equal the per-call
    splits bit for bit, and nothing is kept once the scope is left."""
    from arcnerf_amd.ops import functional as Fn
    from arcnerf_amd.ops.autograd import linear, linear_relu
    from arcnerf_amd.utils.torch_utils import chunk_processing
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(3000, 63, generator=g).cuda()
    w1 = (torch.randn(256, 63, generator=g) / 8).cuda()
    b1 = (torch.randn(256, generator=g) * 0.1).cuda()
    w2 = (torch.randn(256, 256, generator=g) / 16).cuda()
    w3 = (torch.randn(257, 256, generator=g) / 16).cuda()
    up = torch.randn(3000, 257, generator=g).cuda()

    def run(chunk):
        ps = [p.clone().requires_grad_(True) for p in (w1, b1, w2, w3)]
        x = x0.clone().requires_grad_(True)

        def net(xc):
            return linear(linear_relu(linear_relu(xc, ps[0], ps[1]), ps[2], None), ps[3], None)
        y = chunk_processing(net, chunk, False, x)
        (y * up).sum().backward()
        return [y.detach(), x.grad] + [p.grad for p in ps]
    a, b = run(1024), run(0)          # three chunks inside a scope / one call without
    assert Fn._SPLIT_SCOPE is None
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for ga, gb in zip(a[2:], b[2:]):  # (weight gradients: three partial sums against one - summation order, not bits)
        assert (ga - gb).abs().max() <= 2e-6 * gb.abs().max()
    # the same chunks with and without the scope: every bit
    calls = []
    real = Fn.split_weight_scope
    import contextlib
    Fn.split_weight_scope = lambda: (calls.append(1), contextlib.nullcontext())[1]
    try:
        import arcnerf_amd.utils.torch_utils as tu
        c = run(1024)
    finally:
        Fn.split_weight_scope = real
    assert calls, 'chunk_processing did not ask for the scope'
    for ga, gc in zip(a, c):
        assert torch.equal(ga, gc)


@pytest.mark.parametrize('W,D,skips,Wr,Dr,bias,out_act,n,chunk', [
    (256, 8, [4], 128, 1, True, None, 5000, 2048),                    # configs/nerf.yaml, three chunks (the last one ragged)
    (256, 8, [4], 128, 1, True, 'identity', 3000, 0),                 # configs/hdrnerf.yaml: no sigmoid, one chunk
    (128, 4, [1, 2], 64, 2, False, None, 1500, 700),                  # two skips, three radiance layers, no biases, float masks in the radiance net
    (64, 2, [], 32, 1, True, None, 333, 100),                         # no skip, everything on the exact-f32 kernels
    (256, 8, [4], 128, 1, True, None, 2500, 1000)])                   # (n = 2500: 4-D inputs (x / r, 1 / r), the NeRF++ background)
def test_field_chain_equals_the_layer_by_layer_modules(W, D, skips, Wr, Dr, bias, out_act, n, chunk):
    """ops.field_chain.FieldChainFn (GeoNet + RadianceNet 'vf' as one node over all chunks, concatenation-free buffers, weight gradients
    summed by the products) against the same modules evaluated layer by layer under chunk_processing
    (linear_network_module.py:16-335, base_3d_model.py:335-366): outputs to f32 summation order (the permuted layouts move sigma and
    the radiance net's first-layer k-sum to other tiles), every parameter gradient to the order of the chunk sums."""
    from arcnerf_amd.models.base_3d_model import Base3dModel
    from arcnerf_amd.models.base_modules.geo_rad_model.linear_network_module import GeoNet, RadianceNet
    from arcnerf_amd.ops.field_chain import field_chain
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    from arcnerf_amd.utils.torch_utils import chunk_processing
    torch.manual_seed(W + D)
    din = 4 if n == 2500 else 3
    geo = GeoNet(W=W, D=D, skips=skips, encoder=dict_to_obj({'type': 'FreqEmbedder', 'input_dim': din, 'n_freqs': 10}), W_feat=W, use_bias=bias,
                 geometric_init=False).cuda()
    rad = RadianceNet(mode='vf', W=Wr, D=Dr, encoder=dict_to_obj({'view': {'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 4}}), W_feat_in=W,
                      use_bias=bias, out_act_cfg=None if out_act is None else dict_to_obj({'type': out_act})).cuda()
    g = torch.Generator().manual_seed(n)
    pts = (torch.rand(n, din, generator=g) * 4 - 2).cuda()
    dirs = torch.randn(n, 3, generator=g).cuda()
    up_s, up_r = torch.randn(n, generator=g).cuda(), torch.randn(n, 3, generator=g).cuda()
    params = list(geo.parameters()) + list(rad.parameters())

    def grads(sigma, radiance):
        for p in params:
            p.grad = None
        ((sigma * up_s).sum() + (radiance * up_r).sum()).backward()
        return [p.grad.clone() for p in params]
    out = field_chain(geo, rad, pts, dirs, chunk)
    assert out is not None, 'the pair was expected to be eligible'
    got = [o.detach().clone() for o in out], grads(*out)
    ref_out = chunk_processing(Base3dModel._forward_pts_dir, chunk, False, geo, rad, pts, dirs)
    ref = [o.detach().clone() for o in ref_out], grads(*ref_out)
    assert got[0][0].shape == ref[0][0].shape == (n,) and got[0][1].shape == ref[0][1].shape == (n, 3)
    # (sigma is the 257th output there - the split kernel's 256-wide block - and the 257th here, on the exact-f32 remainder kernel)
    assert (got[0][0] - ref[0][0]).abs().max() <= 2e-6 * max(1.0, float(ref[0][0].abs().max()))
    assert (got[0][1] - ref[0][1]).abs().max() <= 2e-6 * max(1.0, float(ref[0][1].abs().max()))
    for a, b in zip(got[1], ref[1]):
        assert a.shape == b.shape
        assert (a - b).abs().max() <= 5e-6 * float(b.abs().max()) + 1e-9
    # without a graph (rendering): the same outputs, nothing kept
    with torch.no_grad():
        s2, r2 = field_chain(geo, rad, pts, dirs, chunk)
    assert torch.equal(s2, got[0][0]) and torch.equal(r2, got[0][1])
    # inputs that want a gradient (normals from a density field) are not the node's business
    assert field_chain(geo, rad, pts.clone().requires_grad_(True), dirs, chunk) is None


@pytest.mark.parametrize('W,D,skips,reduce,norm,wn,bias,n,chunk', [
    (256, 8, [4], True, True, True, True, 3000, 1024),      # configs/neus.yaml: reduced + normalised skip, weight norm, three chunks
    (128, 4, [1], False, False, False, True, 1500, 0),      # plain skip (319-style odd concatenation width), no weight norm, one chunk
    (64, 3, [], False, False, True, False, 700, 300)])      # no skip, no biases, everything on the exact-f32 kernels
def test_sdf_chain_equals_the_double_backward_of_the_modules(W, D, skips, reduce, norm, wn, bias, n, chunk):
    """ops.sdf_chain.SdfChainFn (softplus sdf net + its input gradient as ONE first-order node with a hand-written backward for the
    second-order terms) against GeoNet.forward_with_grad - torch.autograd.grad(create_graph=True) through the layer modules - under a
    loss that uses the sdf, the feature and the normal (an Eikonal term among them), base_network.py:30-44: outputs and the gradient
    of every parameter (weight-norm g and v included)."""
    from arcnerf_amd.models.base_modules.geo_rad_model.linear_network_module import GeoNet
    from arcnerf_amd.ops.sdf_chain import sdf_chain
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    from arcnerf_amd.utils.torch_utils import chunk_processing
    torch.manual_seed(W + D)
    geo = GeoNet(W=W, D=D, skips=skips, encoder=dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 10}), W_feat=W, use_bias=bias,
                 skip_reduce_output=reduce, norm_skip=norm, act_cfg=dict_to_obj({'type': 'softplus', 'beta': 100}), geometric_init=True,
                 radius_init=0.75, weight_norm=wn).cuda()
    g = torch.Generator().manual_seed(n)
    pts = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
    up_f = (torch.randn(n, W, generator=g) * 0.1).cuda()
    up_n = torch.randn(n, 3, generator=g).cuda()
    params = [p for p in geo.parameters()]

    def loss_grads(sdf, feat, normal):
        for p in params:
            p.grad = None
        loss = (sdf[:, 0] ** 2).sum() + (feat * up_f).sum() + ((normal.norm(dim=-1) - 1.0) ** 2).sum() + (normal * up_n).sum()
        loss.backward()
        return [p.grad.clone() for p in params]
    out = sdf_chain(geo, pts, chunk)
    assert out is not None, 'the net was expected to be eligible'
    got = [o.detach().clone() for o in out], loss_grads(*out)
    ref_out = chunk_processing(lambda x: geo.forward_with_grad(x), chunk, False, pts.clone())
    ref = [o.detach().clone() for o in ref_out], loss_grads(*ref_out)
    for a, b, name in zip(got[0], ref[0], ('sdf', 'feature', 'normal')):
        assert a.shape == b.shape, name
        assert (a - b).abs().max() <= 3e-6 * max(1.0, float(b.abs().max())), name
    for a, b, (name, _) in zip(got[1], ref[1], geo.named_parameters()):
        assert a.shape == b.shape, name
        assert (a - b).abs().max() <= 2e-5 * float(b.abs().max()) + 1e-8, (name, float((a - b).abs().max()), float(b.abs().max()))
    with torch.no_grad():          # rendering: normals without a graph
        s2, f2, n2 = sdf_chain(geo, pts, chunk)
    assert torch.equal(n2, got[0][2]) and torch.equal(s2, got[0][0])


@pytest.mark.parametrize('n,chunk', [(1, 0), (1, 4), (9, 4), (130, 128), (257, 1)])
def test_field_and_sdf_nodes_on_tiny_and_ragged_batches(n, chunk):
    """the one-node fields on 1 point, on chunks of 1 and on batches that end in a ragged chunk: outputs and parameter gradients equal
    the module path (chunk_processing of the same nets)"""
    from arcnerf_amd.models.base_3d_model import Base3dModel
    from arcnerf_amd.models.base_modules.geo_rad_model.linear_network_module import GeoNet, RadianceNet
    from arcnerf_amd.ops.field_chain import field_chain
    from arcnerf_amd.ops.sdf_chain import sdf_chain
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    from arcnerf_amd.utils.torch_utils import chunk_processing
    torch.manual_seed(3)
    enc = dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 6})
    geo = GeoNet(W=128, D=3, skips=[1], encoder=enc, W_feat=128, geometric_init=False).cuda()
    rad = RadianceNet(mode='vf', W=128, D=1, encoder=dict_to_obj({'view': {'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 2}}), W_feat_in=128).cuda()
    sdf = GeoNet(W=128, D=3, skips=[1], encoder=enc, W_feat=128, skip_reduce_output=True, norm_skip=True,
                 act_cfg=dict_to_obj({'type': 'softplus', 'beta': 100}), geometric_init=True, weight_norm=True).cuda()
    g = torch.Generator().manual_seed(n)
    pts, dirs = (torch.rand(n, 3, generator=g) - 0.5).cuda(), torch.randn(n, 3, generator=g).cuda()

    def grads(params, loss):
        for p in params:
            p.grad = None
        loss.backward()
        return [p.grad.clone() for p in params]
    ps = list(geo.parameters()) + list(rad.parameters())
    s1, r1 = field_chain(geo, rad, pts, dirs, chunk)
    g1 = grads(ps, s1.sum() + (r1 ** 2).sum())
    s0, r0 = chunk_processing(Base3dModel._forward_pts_dir, chunk, False, geo, rad, pts, dirs)
    g0 = grads(ps, s0.sum() + (r0 ** 2).sum())
    assert s1.shape == s0.shape and r1.shape == r0.shape
    s1, r1, s0, r0 = s1.detach(), r1.detach(), s0.detach(), r0.detach()
    assert (s1 - s0).abs().max() <= 2e-6 * max(1.0, float(s0.abs().max())) and (r1 - r0).abs().max() <= 2e-6
    for a, b in zip(g1, g0):
        assert (a - b).abs().max() <= 1e-5 * float(b.abs().max()) + 1e-9
    ps = list(sdf.parameters())
    d1, f1, n1 = sdf_chain(sdf, pts, chunk)
    g1 = grads(ps, d1.sum() + f1.sum() * 0.01 + ((n1.norm(dim=-1) - 1) ** 2).sum())
    d0, f0, n0 = chunk_processing(lambda x: sdf.forward_with_grad(x), chunk, False, pts.clone())
    g0 = grads(ps, d0.sum() + f0.sum() * 0.01 + ((n0.norm(dim=-1) - 1) ** 2).sum())
    assert d1.shape == d0.shape and f1.shape == f0.shape and n1.shape == n0.shape
    for a, b in ((d1.detach(), d0.detach()), (f1.detach(), f0.detach()), (n1.detach(), n0.detach())):
        assert (a - b).abs().max() <= 3e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(g1, g0):
        assert (a - b).abs().max() <= 3e-5 * float(b.abs().max()) + 1e-8


@pytest.mark.parametrize('ng,frac', [(16, 0.05), (32, 0.6), (64, 0.0), (128, 0.05)])
def test_refresh_cells_and_points_vs_the_torch_selection(ng, frac):
    """arcn_refresh_cells_points against geometry.volume.select_refresh_cells (VolumeBound.optimize, volume_bound.py:178-190): the uniform
    part is the SAME set of n / 4 distinct cells (the image of the seeded bijection), in Z-curve order; the occupied part is the first n / 4 occupied
    cells in flat order; n_valid counts both; every point lies inside its cell's voxel and the jitter is uniform (mean, variance)."""
    from arcnerf_amd.geometry.volume import mix_constants, mix_permutation
    from arcnerf_amd.ops import functional as Fn
    n = ng ** 3
    n_s = n // 4
    g = torch.Generator().manual_seed(ng)
    bf = (torch.rand(n, generator=g) < frac).cuda()
    perm = mix_constants(n, np.random.default_rng(ng))
    cells = torch.full((2 * n_s,), -1, dtype=torch.int64, device='cuda')
    pts = torch.zeros(2 * n_s, 3, device='cuda')
    n_valid = torch.zeros(1, dtype=torch.int32, device='cuda')
    ws = torch.empty(n + 8 * (n // 4096 + 2), dtype=torch.uint8, device='cuda')
    side, mn = 3.0, (-1.5, -1.2, -1.8)
    vs = side / ng
    Fn.refresh_cells_points(bf, ng, perm, vs, mn, 12345, 77, cells, pts, n_valid, ws)
    want_uni = torch.sort(mix_permutation(torch.arange(n_s), n, perm))[0]
    got_uni = cells[:n_s].cpu()
    assert torch.equal(torch.sort(got_uni)[0], want_uni) and want_uni.unique().numel() == n_s
    # ... handed out along the Z-curve: Morton codes strictly increasing
    def spread(v):
        out = torch.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    code = spread(torch.div(got_uni, ng * ng, rounding_mode='floor')) | (spread(torch.div(got_uni, ng, rounding_mode='floor') % ng) << 1) | (spread(got_uni % ng) << 2)
    assert bool((code[1:] > code[:-1]).all())
    occ = torch.nonzero(bf.cpu())[:n_s, 0]
    nv = int(n_valid)
    assert nv == n_s + occ.numel()
    assert torch.equal(cells[n_s:nv].cpu(), occ)
    c = cells[:nv]
    idx3 = torch.stack([torch.div(c, ng * ng, rounding_mode='floor'), torch.div(c, ng, rounding_mode='floor') % ng, c % ng], -1).float()
    lo = idx3 * vs + torch.tensor(mn, device='cuda')
    u = (pts[:nv] - lo) / vs
    assert float(u.min()) >= -1e-4 and float(u.max()) <= 1.0 + 1e-4
    assert abs(float(u.mean()) - 0.5) < 0.01 and abs(float(u.var()) - 1.0 / 12.0) < 0.01
    # another stream position gives another jitter, the same cells
    pts2 = torch.zeros_like(pts)
    Fn.refresh_cells_points(bf, ng, perm, vs, mn, 999, 77, cells, pts2, n_valid, ws)
    assert not torch.equal(pts2[:nv], pts[:nv])


def test_adam_runs_in_one_launch_equal_separate_launches():
    """arcn_adam_ema_step_runs (what is left of the flat buffer beside the levels the scatter's owners update, one launch) against one
    arcn_adam_ema_step per run: parameters, moments, cleared gradients bit for bit; the gaps between the runs untouched"""
    from arcnerf_amd.ops import functional as Fn
    n = 40000
    g = torch.Generator().manual_seed(0)
    base = [torch.randn(n, generator=g).cuda() for _ in range(4)]
    base[3] = base[3].abs()
    runs = [(0, 9826), (12000, 12000 + 4), (20000, 20000 + 7777), (39996, 40000)]
    a = [t.clone() for t in base]
    b = [t.clone() for t in base]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-6, ema_decay=0.95, grad_scale=0.5, zero_grad=True)
    Fn.adam_ema_step_runs(a[0], a[1], a[2], a[3], a[0], runs, 3, **kw)
    for lo, hi in runs:
        Fn.adam_ema_step(b[0][lo:hi], b[1][lo:hi], b[2][lo:hi], b[3][lo:hi], b[0][lo:hi], 3, **kw)
    for x, y, z in zip(a, b, base):
        assert torch.equal(x, y)
        assert torch.equal(x[9826:12000], z[9826:12000]) and torch.equal(x[12004:20000], z[12004:20000])
    assert float(a[1][:9826].abs().max()) == 0.0 and not torch.equal(a[0][:9826], base[0][:9826])


@pytest.mark.parametrize('mode,W,D,wn,bias,pf,n,chunk', [
    ('pvnf', 256, 4, True, True, 0, 3000, 1024),       # configs/neus.yaml: raw positions, weight norm, three chunks
    ('vnf', 128, 2, False, True, 0, 700, 0),           # no position block, one chunk
    ('pf', 64, 2, False, False, 4, 333, 100)])         # encoded positions, no biases, everything on the exact-f32 kernels
def test_radiance_chain_equals_the_layer_by_layer_module(mode, W, D, wn, bias, pf, n, chunk):
    """ops.radiance_chain.RadianceChainFn against RadianceNet.forward under chunk_processing (encoder_mlp_network.py:62-118,
    linear_network_module.py:318-335): radiance, the gradients of every parameter (weight-norm g / v included) and of the normal and
    feature inputs (what NeuS's sdf node differentiates further)."""
    from arcnerf_amd.models.base_modules.geo_rad_model.linear_network_module import RadianceNet
    from arcnerf_amd.ops.radiance_chain import radiance_chain
    from arcnerf_amd.utils.cfgs_utils import dict_to_obj
    from arcnerf_amd.utils.torch_utils import chunk_processing
    torch.manual_seed(W + D)
    enc = dict_to_obj({'pts': {'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': pf}, 'view': {'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 4}})
    Wf = 256 if W == 256 else 64
    rad = RadianceNet(mode=mode, W=W, D=D, encoder=enc, W_feat_in=Wf, use_bias=bias, weight_norm=wn).cuda()
    g = torch.Generator().manual_seed(n)
    x, dirs = (torch.rand(n, 3, generator=g) - 0.5).cuda(), torch.randn(n, 3, generator=g).cuda()
    nrm0, feat0 = torch.randn(n, 3, generator=g).cuda(), (torch.randn(n, Wf + 4, generator=g) * 0.3).cuda()
    up = torch.randn(n, 3, generator=g).cuda()
    params = list(rad.parameters())

    def run(fn):
        nrm = nrm0.clone().requires_grad_(True)
        fbuf = feat0.clone().requires_grad_(True)
        feat = fbuf[:, 1:1 + Wf]              # a column slice of a wider buffer, as the sdf node hands the feature over
        for p in params:
            p.grad = None
        out = fn(x, dirs, nrm, feat)
        (out * up).sum().backward()
        return [out.detach().clone(), nrm.grad.clone() if 'n' in mode else None, fbuf.grad.clone()] + [p.grad.clone() for p in params]
    got = run(lambda a, b, c, d: radiance_chain(rad, a, b, c, d, chunk))
    assert got[0] is not None and got[0].shape == (n, 3)
    ref = run(lambda a, b, c, d: chunk_processing(rad, chunk, False, a, b, c, d))
    assert (got[0] - ref[0]).abs().max() <= 2e-6
    for a, b in zip(got[1:], ref[1:]):
        if a is None and b is None:
            continue
        assert a.shape == b.shape
        assert (a - b).abs().max() <= 1e-5 * float(b.abs().max()) + 1e-9


@pytest.mark.parametrize('act', ['squareplus', 'sine', 'softplus', 'sigmoid'])
def test_activation_table_forward_and_both_derivatives_vs_torch(F, act):
    """The last two activations of the reference's fused-MLP map (tcnn_fusedmlp_module.py:195-213: Squareplus, Sine - tiny-cuda-nn's
    definitions) and two older ones through ops.autograd.ActFn: value, first derivative and the second backward (both outputs of
    arcn_act_bwd_bwd) against torch autograd on the same formula in float64."""
    from arcnerf_amd.ops.autograd import ActFn
    torch.manual_seed(3)
    x = (torch.randn(4096, device='cuda') * 1.5).requires_grad_(True)
    w = torch.randn(4096, device='cuda')
    beta = 1.0

    def ref(t):
        if act == 'squareplus':
            X = 10.0 * t
            return 0.5 * (X + torch.sqrt(X * X + 4.0)) / 10.0
        if act == 'sine':
            return torch.sin(t)
        if act == 'softplus':
            return torch.nn.functional.softplus(t, beta=beta)
        return torch.sigmoid(t)
    y = ActFn.apply(x, act, beta)
    xd = x.detach().double().requires_grad_(True)
    yd = ref(xd)
    assert float((y.double() - yd).abs().max()) < 2e-6
    g, = torch.autograd.grad((y * w).sum(), x, create_graph=True)
    gd, = torch.autograd.grad((yd * w.double()).sum(), xd, create_graph=True)
    assert float((g.double() - gd).abs().max()) < 1e-5
    # second order: differentiate a function of the first gradient wrt x
    h, = torch.autograd.grad((g * g).sum(), x)
    hd, = torch.autograd.grad((gd * gd).sum(), xd)
    assert float((h.double() - hd).abs().max()) <= 2e-5 * max(1.0, float(hd.abs().max()))


def test_fused_mlp_squareplus_trains_and_sine_is_forward_only(F):
    """FusedLayers with tiny-cuda-nn's activation names: Squareplus hidden / output activations against a torch MLP (values, dX, dW);
    Sine evaluates (inference) and its backward is refused with the reason (cos(x) is not a function of the saved sin(x) - tiny-cuda-nn's
    fused MLP has the same restriction)."""
    from arcnerf_amd.models.base_modules.geo_rad_model.tcnn_fusedmlp_module import FusedLayers
    torch.manual_seed(4)
    layers = FusedLayers([32, 64, 16], 'Squareplus', 'Squareplus').cuda()
    x = torch.randn(777, 32, device='cuda', requires_grad=True)
    sq = lambda t: 0.5 * (10.0 * t + torch.sqrt(100.0 * t * t + 4.0)) / 10.0   # noqa: E731
    W0, W1 = layers.params[:32 * 64].view(64, 32), layers.params[32 * 64:].view(16, 64)
    y = layers(x)
    ref = sq(sq(x @ W0.t()) @ W1.t())
    assert float((y - ref).abs().max()) < 1e-5
    g = torch.randn_like(y)
    dx, dw = torch.autograd.grad((y * g).sum(), [x, layers.params])
    rdx, rdw = torch.autograd.grad((ref * g).sum(), [x, layers.params])
    assert float((dx - rdx).abs().max()) <= 1e-4 * float(rdx.abs().max()) and float((dw - rdw).abs().max()) <= 1e-4 * float(rdw.abs().max())
    sine = FusedLayers([32, 64, 16], 'Sine', 'None').cuda()
    with torch.no_grad():
        ys = sine(x)
        V0, V1 = sine.params[:32 * 64].view(64, 32), sine.params[32 * 64:].view(16, 64)
        assert float((ys - torch.sin(x @ V0.t()) @ V1.t()).abs().max()) < 1e-5
    with pytest.raises(RuntimeError, match='Sine'):
        sine(x).sum().backward()


def test_imgloss_huber_fused_equals_the_elementwise_form(F):
    """trainer.ImgLoss (Huber, plain mean) takes value and gradient from one kernel on CUDA tensors (ops.autograd.HuberMeanFn); the
    masked / weighted forms keep the reference's elementwise chain (loss/img_loss.py:60-100).  Same loss and gradient."""
    from arcnerf_amd import trainer as T
    torch.manual_seed(2)
    x = torch.rand(1, 777, 3, device='cuda', requires_grad=True)
    y = torch.rand(1, 777, 3, device='cuda')
    cfg = type('C', (), dict(keys=['rgb_coarse'], loss_type='Huber', delta=0.1, weight=3000.0))()
    fused = T.ImgLoss(cfg)({'img': y}, {'rgb_coarse': x})
    gf, = torch.autograd.grad(fused * 3000.0, x)
    a = (x - y).abs()
    ref = torch.where(a < 0.1, 0.5 / 0.1 * a ** 2, a - 0.05).mean()
    gr, = torch.autograd.grad(ref * 3000.0, x)
    assert abs(float(fused) - float(ref)) <= 1e-6 * float(ref) and float((gf - gr).abs().max()) <= 1e-5 * float(gr.abs().max())


@pytest.mark.gpu
def test_persistent_marcher_gives_the_samples_of_the_one_wave_per_ray_marcher():
    """arcn_march_count_waves (wave w marches the rays w, w + n_waves, ...: the launch of a batch marched two steps ahead) against
    arcn_march_count_culled: counts, bounds and every emitted t bit for bit, with and without the culling grid, for wave counts that divide
    the batch, do not divide it, exceed it (falls back to one wave per ray) and for a batch with a ragged last workgroup"""
    import numpy as np
    from arcnerf_amd import _native as N
    from arcnerf_amd.pipeline import NgpConfig, NgpField, NgpPipeline, synthetic_bitfield, synthetic_rays
    dev = torch.device('cuda:0')
    cfg = NgpConfig()
    fld = NgpField(cfg, device=dev, seed=0)
    pipe = NgpPipeline(fld, max_rays=32768, max_samples=1 << 19, packed_bits=True)
    pipe.set_bitfield(torch.from_numpy(synthetic_bitfield(cfg.n_grid, 0.05, seed=3)))
    L = N.lib()
    for R in (4096, 1023, 32768):      # (32768 rays x 4096 waves: the headline's launch at the largest batch of the dynamic batch size)
        o, d = synthetic_rays(R, seed=41, device=dev)

        def march(waves, coarse):
            out = {k: torch.zeros_like(pipe.buf[k]) for k in ('scratch_t', 'counts', 'near', 'far')}
            out['scratch_t'].fill_(-1.0)
            N.check(L.arcn_march_count_waves(N.ptr(o), N.ptr(d), N.ptr(pipe.aabb23), cfg.n_grid, N.ptr(pipe._occ()), int(pipe.packed_bits),
                                             N.ptr(coarse), cfg.n_sample, cfg.dt, cfg.near_distance, 0, 12345, 67, N.ptr(out['scratch_t']),
                                             N.ptr(out['counts']), N.ptr(out['near']), N.ptr(out['far']), R, waves, N.stream()), 'march_count_waves')
            torch.cuda.synchronize()
            return out
        ref = {k: torch.zeros_like(pipe.buf[k]) for k in ('scratch_t', 'counts', 'near', 'far')}
        ref['scratch_t'].fill_(-1.0)
        N.check(L.arcn_march_count_culled(N.ptr(o), N.ptr(d), N.ptr(pipe.aabb23), cfg.n_grid, N.ptr(pipe._occ()), int(pipe.packed_bits),
                                          N.ptr(pipe._coarse), cfg.n_sample, cfg.dt, cfg.near_distance, 0, 12345, 67, N.ptr(ref['scratch_t']),
                                          N.ptr(ref['counts']), N.ptr(ref['near']), N.ptr(ref['far']), R, N.stream()), 'march_count_culled')
        torch.cuda.synchronize()
        cnt = ref['counts'][:R].cpu().numpy()
        assert cnt.sum() > 10 * R and (cnt == 0).mean() > 0.3
        for waves in ((0, 64, 256, 1000, 2048, 100000) if R <= 4096 else (4096, 5000)):
            for coarse in (pipe._coarse, None):
                got = march(waves, coarse)
                assert torch.equal(got['counts'][:R], ref['counts'][:R]), (R, waves)
                assert torch.equal(got['near'][:R], ref['near'][:R]) and torch.equal(got['far'][:R], ref['far'][:R])
                t_got, t_ref = got['scratch_t'][:R].cpu().numpy(), ref['scratch_t'][:R].cpu().numpy()
                valid = np.arange(t_ref.shape[1])[None, :] < cnt[:, None]
                assert np.array_equal(t_got[valid], t_ref[valid]), (R, waves)


@pytest.mark.gpu
def test_dense_layers_refuse_non_float32_cuda_tensors_instead_of_the_library():
    """the product path has no library fallback: a CUDA tensor that is not float32 used to slip to torch.nn.functional.linear (hipBLASLt)
    silently - a dtype slip would have benchmarked the library.  Every dense-layer front end raises instead (the extension of
    tests/test_host_api.py::test_product_path_has_no_cpu_fallback to the dtype case); float32 takes the hand-written products."""
    from arcnerf_amd.ops import autograd as A
    x, w = torch.randn(64, 32, device='cuda'), torch.randn(16, 32, device='cuda')
    for bad_x, bad_w in ((x.half(), w.half()), (x.double(), w.double()), (x.bfloat16(), w), (x, w.half())):
        for fn in (lambda: A.linear(bad_x, bad_w), lambda: A.linear_relu(bad_x, bad_w), lambda: A.linear_softplus(bad_x, bad_w, None, 100.0),
                   lambda: A.linear_act_nograd(bad_x, bad_w, None, 'relu')):
            with pytest.raises(RuntimeError, match='float32'):
                fn()
    with pytest.raises(RuntimeError, match='float32'):
        A.softplus(x.double(), 100.0)
    y = A.linear(x, w)
    assert torch.allclose(y, x @ w.t(), rtol=1e-5, atol=1e-5) and A._use_hip_linear(x, w)
    # CPU tensors (the host-side tests of the module logic) still go to torch
    assert not A._use_hip_linear(x.cpu(), w.cpu())
