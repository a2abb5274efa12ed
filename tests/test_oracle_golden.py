"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference's own torch path.

Tolerances: integer outputs (voxel / corner / hash rows / searchsorted bins / masks) bit-exact;
fp32 outputs 1e-6 relative-ish (summation order may differ from torch's vectorised kernels).
"""
import numpy as np
import pytest

from conftest import load_golden, make_table


def close(a, b, rtol=2e-6, atol=2e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---- G11 pcg32: published known-answer vector of the PCG reference implementation ---------------
def test_pcg32_known_answer(oracle):
    # pcg32-demo (pcg-c-basic), pcg32_srandom(42u, 54u), round 1
    r = oracle.Pcg32(42, 54)
    got = r.next_uint(6)
    want = np.array([0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e], dtype=np.uint32)
    assert (got == want).all()


def test_pcg32_advance_matches_stepping(oracle):
    a = oracle.Pcg32(9121)
    b = a.copy()
    a.next_uint(8 * 37)
    b.advance(8 * 37)
    assert a.state == b.state
    f = a.next_float(16)
    assert (f >= 0).all() and (f < 1).all()
    c = oracle.Pcg32(9121)
    c.advance()  # 2^32
    d = oracle.Pcg32(9121)
    d.advance(1 << 31)
    d.advance(1 << 31)
    assert c.state == d.state
    # going back
    c.advance(-(1 << 32))
    assert c.state == oracle.Pcg32(9121).state


# ---- G1 compositing ------------------------------------------------------------------------------
@pytest.mark.parametrize('P', [2, 17, 64])
def test_g1_ray_marching(oracle, P):
    g = load_golden('g1_compositing')
    pre = 'P{}_'.format(P)
    sigma, rad, z = g[pre + 'sigma'], g[pre + 'radiance'], g[pre + 'zvals']
    for add_inf_z in (False, True):
        for mode in ('none', 'white', 'bkg_full', 'bkg_one'):
            tag = 'P{}_inf{}_{}_'.format(P, int(add_inf_z), mode)
            bkg = g[pre + mode] if mode.startswith('bkg') else None
            out = oracle.ray_marching_fwd(sigma, rad, z, add_inf_z=add_inf_z, white_bkg=(mode == 'white'), bkg_color=bkg)
            for k in ('rgb', 'depth', 'mask', 'alpha', 'trans_shift', 'weights'):
                close(out[k], g[tag + k], rtol=1e-5, atol=1e-6)
            d_sigma, d_rad = oracle.ray_marching_bwd(
                sigma, rad, z, g[pre + 'g_rgb'], g[pre + 'g_depth'], g[pre + 'g_mask'], add_inf_z=add_inf_z,
                white_bkg=(mode == 'white'), bkg_color=bkg)
            close(d_rad, g[tag + 'd_radiance'], rtol=1e-5, atol=1e-6)
            close(d_sigma, g[tag + 'd_sigma'], rtol=2e-4, atol=2e-5)
    # alpha branch
    tag = 'P{}_alpha_'.format(P)
    out = oracle.ray_marching_fwd(None, rad, z, alpha=g[tag + 'in'], bkg_color=g[pre + 'bkg_full'])
    for k in ('rgb', 'depth', 'mask', 'trans_shift', 'weights'):
        close(out[k], g[tag + k], rtol=1e-5, atol=1e-6)
    d_alpha, d_rad = oracle.ray_marching_bwd(None, rad, z, g[pre + 'g_rgb'], g[pre + 'g_depth'], g[pre + 'g_mask'],
                                             alpha=g[tag + 'in'], bkg_color=g[pre + 'bkg_full'])
    close(d_rad, g[tag + 'd_radiance'], rtol=1e-5, atol=1e-6)
    close(d_alpha, g[tag + 'd_alpha'], rtol=2e-4, atol=2e-5)


def test_ray_marching_rejects_decreasing_z(oracle):
    z = np.array([[1.0, 0.5, 2.0]], np.float32)
    with pytest.raises(AssertionError):
        oracle.ray_marching_fwd(np.ones((1, 3), np.float32), np.ones((1, 3, 3), np.float32), z)


# ---- G2 resampling -------------------------------------------------------------------------------
def test_g2_resampling(oracle):
    g = load_golden('g2_resampling')
    # integer bins bit-exact given the reference's cdf
    s, inds = oracle.sample_cdf(g['bins'], g['cdf'], g['u_det'])
    assert (inds == g['inds_det']).all()
    close(s, g['samples_det'], rtol=1e-6, atol=1e-6)
    s, inds = oracle.sample_cdf(g['bins'], g['cdf'], g['u'])
    assert (inds == g['inds_rnd']).all()
    close(s, g['samples_rnd'], rtol=1e-6, atol=1e-6)
    # full sample_pdf (own cdf).  The cumsum is restated exactly (double running sum, every prefix rounded to float: torch's CPU kernel);
    # the normaliser is not (torch.sum's vectorised float order depends on the host ISA), so the cdf agrees to the normaliser's last
    # ulp or two - and bit for bit on the rows where the two normalisers coincide
    own_cdf = oracle.weights_to_cdf(g['weights'])
    close(own_cdf, g['cdf'], rtol=3e-7, atol=1.5e-7)
    w = g['weights'] + np.float32(1e-5)
    exact_rows = (own_cdf == g['cdf']).all(1)
    assert exact_rows.mean() > 0.3
    # (the `denom < eps -> 1` rule of sample_cdf is discontinuous, so a last-ulp cdf difference may move a
    #  handful of samples inside a near-empty bin: bound their count instead of their value)
    own = oracle.sample_pdf(g['bins'], g['weights'], 128)
    bad = np.abs(own - g['samples_det']) > 1e-4
    assert bad.mean() < 2e-3


# ---- G3 zvals ------------------------------------------------------------------------------------
def test_g3_zvals(oracle):
    g = load_golden('g3_zvals')
    for n_pts in (2, 64, 129):
        for inc in (True, False):
            for inv in (True, False):
                z = oracle.zvals_from_near_far(g['near'], g['far'], n_pts, inclusive=inc, inverse_linear=inv)
                close(z, g['n{}_inc{}_inv{}'.format(n_pts, int(inc), int(inv))], rtol=1e-6, atol=1e-6)


# ---- G4 intersections ----------------------------------------------------------------------------
def test_g4_aabb_torch_semantics(oracle):
    g = load_golden('g4_intersections')
    near, far, pts, mask = oracle.aabb_intersection_torch(g['rays_o'], g['rays_d'], g['aabb'])
    assert (mask == g['mask']).all()
    close(near, g['near'])
    close(far, g['far'])
    close(pts, g['pts'], atol=1e-5)


def test_g4_sphere_intersection(oracle):
    """sphere_ray_intersection (geometry/ray.py:180-255) incl. the hand cases of tests_ray.py (outside / tangent / inside /
    on-surface rays): masks identical, near / far / points to 1 ulp (torch's dot-product summation order)."""
    g = load_golden('g4_intersections')
    near, far, pts, mask = oracle.sphere_intersection(g['rays_o'], g['rays_d'], g['radius'])
    assert (mask == g['s_mask']).all()
    close(near, g['s_near'], rtol=1e-6, atol=1e-6)
    close(far, g['s_far'], rtol=1e-6, atol=1e-6)
    close(pts, g['s_pts'], rtol=1e-6, atol=2e-6)
    assert (near[~mask] == 0).all() and (far[~mask] == 0).all() and (near >= 0).all() and (far >= near).all()
    inside = np.linalg.norm(g['rays_o'], axis=-1) < g['radius'][0] - 1e-3
    assert mask[inside, 0].all() and (near[inside, 0] == 0).all()


def test_k2_consistent_with_torch_path(oracle):
    """K2 (CUDA semantics) has no runnable reference: where it reports a hit with tmin>0 it must agree with
    the torch path up to the torch path's +-eps shift."""
    g = load_golden('g4_intersections')
    aabb23 = np.ascontiguousarray(np.transpose(g['aabb'], (0, 2, 1)))
    near, far, pts, mask = oracle.aabb_intersection(g['rays_o'], g['rays_d'], aabb23)
    both = mask & g['mask'] & (g['near'] > 1e-6)
    assert both.sum() > 100
    close(near[both], g['near'][both], atol=1e-5)
    close(far[both], g['far'][both], atol=1e-5)
    # K2 masks out rays that start inside the box (tmin <= 0)
    inside = (np.abs(g['rays_o']) < 1.0).all(-1)
    assert not mask[inside, 0].any()


# ---- G5 voxel math -------------------------------------------------------------------------------
@pytest.mark.parametrize('n_grid', [8, 15, 128])
def test_g5_voxel_info(oracle, n_grid):
    g = load_golden('g5_voxel')
    mn, mx = np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32)
    vidx, valid, cidx, w = oracle.voxel_grid_info(g['pts'], mn, mx, n_grid)
    assert (valid == g['n{}_valid'.format(n_grid)]).all()
    assert (vidx == g['n{}_voxel_idx'.format(n_grid)]).all()
    assert (cidx[valid] == g['n{}_corner_idx'.format(n_grid)]).all()
    close(w[valid], g['n{}_weights'.format(n_grid)], rtol=1e-6, atol=1e-7)


def test_g5_occupancy_lookup(oracle):
    g = load_golden('g5_voxel')
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    occ = oracle.check_pts_in_occ_voxel(g['pts'], g['n8_bitfield'], aabb23, 8)
    assert (occ == g['n8_pts_in_occ']).all()


# ---- G6 hash grid --------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', ['ngp', 'tiny', 'f4'])
def test_g6_hashgrid(oracle, tag):
    g = load_golden('g6_hashgrid')
    L, F, T, base, mx_res = [int(v) for v in g[tag + '_cfg']]
    res, offs = oracle.hashgrid_levels(L, T, base, mx_res)
    assert (res == g[tag + '_resolutions']).all()
    assert (offs == g[tag + '_offsets']).all()
    if tag == 'ngp':
        assert list(res) == [15, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
        assert offs[-1] == 6098108
    table = make_table(int(offs[-1]), F, seed=7, scale=0.5)
    mn, mx = np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32)
    out, idx = oracle.hashgrid_fwd(g[tag + '_xyz'], table, res, offs, mn, mx, with_idx=True)
    assert (idx == g[tag + '_hash_idx']).all()  # integer hash rows: bit exact
    close(out, g[tag + '_out'], rtol=1e-5, atol=1e-6)
    dtable, dxyz = oracle.hashgrid_bwd(g[tag + '_xyz'], table, g[tag + '_g_out'], res, offs, mn, mx, want_dxyz=True)
    rows = g[tag + '_d_table_rows']
    nz = np.nonzero(np.abs(dtable).sum(-1) > 0)[0]
    assert set(nz.tolist()) <= set(rows.tolist())
    close(dtable[rows], g[tag + '_d_table_vals'], rtol=1e-4, atol=1e-5)
    scale = np.abs(g[tag + '_d_xyz']).max()
    close(dxyz, g[tag + '_d_xyz'], rtol=1e-4, atol=1e-4 * scale)


@pytest.mark.parametrize('tag', ['ngp', 'tiny', 'f4'])
def test_g17_hashgrid_second_order(oracle, tag):
    """the encoding's input gradient differentiated again, against the reference's torch backend under double autograd"""
    g = load_golden('g17_hashgrid_second_order')
    L, F, T, base, mx_res = [int(v) for v in g[tag + '_cfg']]
    res, offs = oracle.hashgrid_levels(L, T, base, mx_res)
    table = make_table(int(offs[-1]), F, seed=17, scale=0.5)
    mn, mx = np.full(3, -1.0, np.float32), np.full(3, 1.0, np.float32)
    _, dx = oracle.hashgrid_bwd(g[tag + '_xyz'], table, g[tag + '_gy'], res, offs, mn, mx, want_dxyz=True)
    close(dx, g[tag + '_dx'], rtol=1e-4, atol=1e-4 * np.abs(g[tag + '_dx']).max())
    ddout, dtable, d2x = oracle.hashgrid_bwd_bwd(g[tag + '_xyz'], g[tag + '_gdx'], table, g[tag + '_gy'], res, offs, mn, mx)
    close(ddout, g[tag + '_d_gy'], rtol=1e-4, atol=1e-4 * np.abs(g[tag + '_d_gy']).max())
    rows = g[tag + '_d_table_rows']
    nz = np.nonzero(np.abs(dtable).sum(-1) > 0)[0]
    assert set(nz.tolist()) <= set(rows.tolist())
    close(dtable[rows], g[tag + '_d_table_vals'], rtol=1e-4, atol=1e-4 * np.abs(g[tag + '_d_table_vals']).max())
    close(d2x, g[tag + '_d_x'], rtol=1e-4, atol=2e-4 * np.abs(g[tag + '_d_x']).max())


def test_g20_get_rays(oracle):
    g = load_golden('g20_get_rays')
    W, H, K, c2w, idx = int(g['W']), int(g['H']), g['K'], g['c2w'], g['index']
    cases = {'wh': dict(wh_order=True), 'hw': dict(wh_order=False), 'center': dict(wh_order=True, center_pixel=True),
             'raw': dict(wh_order=False, normalize_rays_d=False), 'ndc': dict(wh_order=True, ndc=True, ndc_near=1.0),
             'idx': dict(index=idx), 'idx_center_ndc': dict(index=idx, center_pixel=True, ndc=True, ndc_near=0.5)}
    for tag, kw in cases.items():
        o, d, r = oracle.get_rays(W, H, K, c2w, **kw)
        close(o, g[tag + '_o'], rtol=1e-5, atol=1e-5)
        close(d, g[tag + '_d'], rtol=1e-5, atol=1e-5)
        if tag + '_r' in g.files:
            close(r, g[tag + '_r'], rtol=1e-4, atol=1e-6)
        else:
            assert r is None and (g[tag + '_flat'] == idx[:, 0] * H + idx[:, 1]).all()


# ---- G7 freq / SH --------------------------------------------------------------------------------
def test_g7_freq(oracle):
    g = load_golden('g7_freq_sh')
    for n_freqs in (10, 4, 0):
        for inc in (True, False):
            if n_freqs == 0 and not inc:
                continue
            t = 'freq{}_inc{}'.format(n_freqs, int(inc))
            # 2^9 * x amplifies the 1-ulp argument difference of sinf implementations
            close(oracle.freq_fwd(g['x'], n_freqs, inc), g[t], rtol=0, atol=2e-6)
            dx = oracle.freq_bwd(g['x'], g[t + '_g'], n_freqs, inc)
            close(dx, g[t + '_dx'], rtol=1e-5, atol=1e-3)


def test_g7_sh(oracle):
    g = load_golden('g7_freq_sh')
    for deg in (1, 2, 3, 4, 5):
        for inc in (True, False):
            close(oracle.sh_fwd(g['dirs'], deg, inc), g['sh{}_inc{}'.format(deg, int(inc))], rtol=1e-6, atol=1e-6)


# ---- G8 MLPs -------------------------------------------------------------------------------------
def _geo_fwd(oracle, g, t, x, bias):
    W0, W1 = g[t + '.layers.0.0.weight'] if (t + '.layers.0.0.weight') in g else g[t + '.layers.0.weight'], None
    return W0, W1


def _layer_keys(g, prefix):
    """(weight, bias|None) per layer, in order, from an exported GeoNet/RadianceNet state_dict."""
    ws = sorted([k for k in g.files if k.startswith(prefix + '.layers.') and k.endswith('weight')],
                key=lambda k: int(k[len(prefix) + 8:].split('.')[0]))
    out = []
    for k in ws:
        b = k[:-6] + 'bias'
        out.append((g[k], g[b] if b in g.files else None))
    return out


@pytest.mark.parametrize('bias', [0, 1])
def test_g8_ngp_geo_and_radiance(oracle, bias):
    g = load_golden('g8_mlps')
    t = 'geo_b{}'.format(bias)
    (W0, b0), (W1, b1) = _layer_keys(g, t)
    assert W0.shape == (64, 32) and W1.shape == (16, 64)
    h, pre_h = oracle.linear_fwd(g[t + '_x'], W0, b0, 'relu', want_pre=True)
    o = oracle.linear_fwd(h, W1, b1, None)
    sigma = oracle.act_fwd(o[:, :1], 'truncexp')
    close(sigma, g[t + '_sigma'], rtol=1e-5, atol=1e-6)
    close(o[:, 1:], g[t + '_feat'], rtol=1e-5, atol=1e-6)
    # backward
    do = np.concatenate([oracle.act_bwd(o[:, :1], sigma, g[t + '_g_sigma'], 'truncexp'), g[t + '_g_feat']], 1)
    dh, dW1, db1 = oracle.linear_bwd(h, W1, o, o, do, None, has_bias=bool(bias))
    dx, dW0, db0 = oracle.linear_bwd(g[t + '_x'], W0, pre_h, h, dh, 'relu', has_bias=bool(bias))
    names = [k for k in g.files if k.startswith(t + '_grad.') and k.endswith('weight')]
    names.sort()
    close(dW0, g[names[0]], rtol=1e-4, atol=1e-4)
    close(dW1, g[names[1]], rtol=1e-4, atol=1e-4)
    close(dx, g[t + '_dx'], rtol=1e-4, atol=1e-5)

    t = 'rad_b{}'.format(bias)
    layers = _layer_keys(g, t)
    assert [w.shape for w, _ in layers] == [(64, 32), (64, 64), (3, 64)]
    v = g[t + '_view']
    v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    x = np.concatenate([g[t + '_feat'], oracle.sh_fwd(v, 4, False)], 1)  # mode 'fv': feat first
    for i, (W, b) in enumerate(layers):
        x = oracle.linear_fwd(x, W, b, 'relu' if i < 2 else 'sigmoid')
    close(x, g[t + '_rgb'], rtol=1e-5, atol=1e-6)


def test_g8_skip_net_and_truncexp(oracle):
    g = load_golden('g8_mlps')
    layers = _layer_keys(g, 'geo_skip')
    emb = oracle.freq_fwd(g['geo_skip_x'], 10, True)
    h = emb
    for i, (W, b) in enumerate(layers):
        h = oracle.linear_fwd(h, W, b, 'relu' if i < len(layers) - 1 else None)
        if i == 2:
            h = np.concatenate([h, emb], 1)  # skips=[2]: [h, x_embed]
    close(h[:, :1], g['geo_skip_sigma'], rtol=1e-4, atol=1e-5)
    close(h[:, 1:], g['geo_skip_feat'], rtol=1e-4, atol=1e-5)
    # 'vf' order: view embedding first, then feat
    layers = _layer_keys(g, 'rad_vf')
    v = g['rad_vf_view']
    v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    x = np.concatenate([oracle.freq_fwd(v, 4, True), g['geo_skip_feat']], 1)
    for i, (W, b) in enumerate(layers):
        x = oracle.linear_fwd(x, W, b, 'relu' if i < len(layers) - 1 else 'sigmoid')
    close(x, g['rad_vf_rgb'], rtol=1e-4, atol=1e-5)
    y = oracle.act_fwd(g['truncexp_x'], 'truncexp')
    close(y, g['truncexp_y'], rtol=1e-6, atol=0)
    close(oracle.act_bwd(g['truncexp_x'], y, np.ones_like(y), 'truncexp'), g['truncexp_dx'], rtol=1e-6, atol=0)


# ---- G10 occupancy update ------------------------------------------------------------------------
def test_g10_occupancy_update(oracle):
    g = load_golden('g10_occupancy')
    opa = np.ascontiguousarray(g['opa0'].reshape(-1).copy())
    oracle.update_opafield(opa, g['flat_idx'], g['new_opacity'], ema=0.95)
    close(opa.reshape(8, 8, 8), g['opa1'], rtol=0, atol=0)
    bf, thres = oracle.update_bitfield_by_opafield(opa.reshape(8, 8, 8), 0.01)
    assert (bf == g['bitfield']).all()
    assert abs(thres - min(float(g['mean_opa']), 0.01)) < 1e-8


# ---- K3 sampler: invariants from the torch side (no runnable reference for the CUDA kernel) ------
def test_k3_sampler_invariants(oracle):
    rng = np.random.default_rng(5)
    n_grid, n_pts = 16, 256
    bf = rng.random((n_grid, n_grid, n_grid)) < 0.15
    R = 300
    o = rng.normal(size=(R, 3)).astype(np.float32)
    o = o / np.linalg.norm(o, axis=-1, keepdims=True) * 2.5
    tgt = (rng.random((R, 3)).astype(np.float32) - 0.5) * 1.6
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    aabb23 = np.array([[-1, -1, -1], [1, 1, 1]], np.float32)
    near, far, _, mask_r = oracle.aabb_intersection(o, d, aabb23[None])
    dt = np.float32(2 * np.sqrt(3.0) / n_pts)
    host = oracle.Pcg32(9121)
    z, m, cnt, trace = oracle.sparse_volume_sampling(o, d, near, far, n_pts, dt, aabb23, n_grid, bf, 0.2, host.state,
                                                     host.inc, with_trace=True)
    assert (m.sum(1) == cnt).all()
    # mask rows are [T..T F..F]; padded zvals repeat the last valid z; empty rows are all zero
    for r in range(R):
        c = cnt[r]
        assert m[r, :c].all() and not m[r, c:].any()
        if c == 0:
            assert (z[r] == 0).all()
        else:
            assert (z[r, c:] == z[r, c - 1]).all()
            assert (np.diff(z[r, :c]) > 0).all()
            assert z[r, 0] >= max(near[r, 0], 0.2) and z[r, c - 1] <= far[r, 0]
    # every emitted sample is in an occupied voxel (Volume.check_pts_in_occ_voxel semantics) and the traced
    # flat index is that voxel
    rr, jj = np.nonzero(m)
    pts = o[rr] + d[rr] * z[rr, jj][:, None]
    assert oracle.check_pts_in_occ_voxel(pts, bf, aabb23, n_grid).all()
    vidx, valid, _, _ = oracle.voxel_grid_info(pts, aabb23[0], aabb23[1], n_grid)
    flat = vidx[:, 0] * n_grid * n_grid + vidx[:, 1] * n_grid + vidx[:, 2]
    assert valid.all() and (flat == trace[rr, jj]).all()
    assert cnt.sum() > 1000
    # the jitter of ray i is draw number i*8 of the host stream
    h2 = oracle.Pcg32(9121)
    h2.advance(8 * 7)
    u7 = h2.next_float(1)[0]
    start7 = np.float32(max(near[7, 0], 0.2)) + dt * u7
    if cnt[7] > 0:
        k = np.round((z[7, 0] - start7) / dt)
        assert abs(z[7, 0] - (start7 + k * dt)) < 1e-4


# ---- G12 NeuS interval opacity -----------------------------------------------------------------------
@pytest.mark.parametrize('tag,s', [('s64', 64.0), ('s4', 4.0), ('s800', 800.0)])
@pytest.mark.parametrize('clip', [True, False])
def test_g12_sdf_to_alpha(oracle, tag, s, clip):
    """sdf_to_alpha fwd + gradients w.r.t. mid sdf / slope / the scale against the reference's autograd (neus_model.py:242-265)."""
    g = load_golden('g12_neus')
    key = '{}_clip{}'.format(tag, int(clip))
    alpha = oracle.sdf_to_alpha_fwd(g['mid_sdf'], g['zvals'], g['mid_slope'], s, clip)
    close(alpha, g[key + '_alpha'], rtol=1e-6, atol=5e-7)
    d_sdf, d_slope, d_s = oracle.sdf_to_alpha_bwd(g['mid_sdf'], g['zvals'], g['mid_slope'], s, g[key + '_gout'], clip)
    close(d_sdf, g[key + '_d_sdf'], rtol=1e-4, atol=1e-5 * np.abs(g[key + '_d_sdf']).max())
    close(d_slope, g[key + '_d_slope'], rtol=1e-4, atol=1e-5 * max(np.abs(g[key + '_d_slope']).max(), 1e-30))
    assert abs(d_s - float(g[key + '_d_s'])) <= 1e-4 * abs(float(g[key + '_d_s'])) + 1e-7
    dup = g['zvals'][:, 1:] == g['zvals'][:, :-1]          # padded tails: zero-length intervals carry alpha = 1e-5 / (cdf + 1e-5)
    assert dup.any() and (d_slope[dup] == 0).all()
