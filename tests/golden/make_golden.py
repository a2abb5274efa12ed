"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE's torch path.

Run only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The .npz files hold inputs and the reference's outputs (data only).  Nothing here travels as
code to the GPU box; tests read the .npz files and never import the reference.
Vector ids follow SURVEY.md §8(c): G1 compositing, G2 resampling, G3 zvals, G4 intersections,
G5 voxel math, G6 hash grid, G7 freq/SH, G8 MLPs + TruncExp, G10 occupancy update, G12 NeuS interval opacity.
(G9 end-to-end model vectors are made by make_golden_models.py; G11 pcg32 is a published
known-answer vector checked in tests/test_oracle_golden.py.)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)

# the only non-importable dependency of the hot path modules: 4 helper names never called on the path
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

from arcnerf.render import ray_helper  # noqa: E402
from arcnerf.geometry.ray import aabb_ray_intersection, sphere_ray_intersection  # noqa: E402
from arcnerf.geometry.volume import Volume  # noqa: E402
from arcnerf.models.base_modules.encoding.hashgrid_encoder import HashGridEmbedder  # noqa: E402
from arcnerf.models.base_modules.encoding.freq_encoder import FreqEmbedder  # noqa: E402
from arcnerf.models.base_modules.encoding.sh_encoder import SHEmbedder  # noqa: E402
from arcnerf.models.base_modules.geo_rad_model.linear_network_module import GeoNet, RadianceNet  # noqa: E402
from arcnerf.ops.trunc_exp import TruncExp  # noqa: E402
from common.utils.cfgs_utils import dict_to_obj  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npy(t):
    return None if t is None else t.detach().cpu().numpy()


def save(name, **arrs):
    arrs = {k: v for k, v in arrs.items() if v is not None}
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('{:28s} {:8.1f} KB'.format(name, os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------------
def g1_compositing():
    g = torch.Generator().manual_seed(101)
    R = 12
    cases = {}
    for P in (2, 17, 64):
        sigma = torch.rand(R, P, generator=g) * 30.0
        sigma[torch.rand(R, P, generator=g) < 0.3] = 0.0  # zero runs
        sigma[:, P // 2:] *= (torch.rand(R, 1, generator=g) > 0.5).float()
        sigma = sigma - 2.0 * (torch.rand(R, P, generator=g) < 0.1).float()  # some negatives (relu)
        rad = torch.rand(R, P, 3, generator=g)
        z = torch.sort(torch.rand(R, P, generator=g) * 3.0 + 0.5, dim=-1)[0]
        # duplicate-z tails on half of the rays (masked-sample convention)
        for r in range(0, R, 2):
            k = max(1, int(torch.randint(1, P + 1, (1,), generator=g)))
            z[r, k - 1:] = z[r, k - 1]
        bkg_full = torch.rand(R, 3, generator=g)
        bkg_one = torch.rand(1, 3, generator=g)
        g_rgb, g_d, g_m = torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g)
        cases['P{}_sigma'.format(P)] = npy(sigma)
        cases['P{}_radiance'.format(P)] = npy(rad)
        cases['P{}_zvals'.format(P)] = npy(z)
        cases['P{}_bkg_full'.format(P)] = npy(bkg_full)
        cases['P{}_bkg_one'.format(P)] = npy(bkg_one)
        cases['P{}_g_rgb'.format(P)] = npy(g_rgb)
        cases['P{}_g_depth'.format(P)] = npy(g_d)
        cases['P{}_g_mask'.format(P)] = npy(g_m)
        for add_inf_z in (False, True):
            for mode in ('none', 'white', 'bkg_full', 'bkg_one'):
                s = sigma.clone().requires_grad_(True)
                c = rad.clone().requires_grad_(True)
                out = ray_helper.ray_marching(
                    s, c, z.clone(), add_inf_z=add_inf_z, white_bkg=(mode == 'white'),
                    bkg_color={'bkg_full': bkg_full, 'bkg_one': bkg_one}.get(mode)
                )
                loss = (out['rgb'] * g_rgb).sum() + (out['depth'] * g_d).sum() + (out['mask'] * g_m).sum()
                loss.backward()
                tag = 'P{}_inf{}_{}'.format(P, int(add_inf_z), mode)
                for k in ('rgb', 'depth', 'mask', 'alpha', 'trans_shift', 'weights'):
                    cases[tag + '_' + k] = npy(out[k])
                cases[tag + '_d_sigma'] = npy(s.grad)
                cases[tag + '_d_radiance'] = npy(c.grad)
        # alpha= branch (NeuS), sigma None
        a = (torch.rand(R, P, generator=g)).requires_grad_(True)
        c = rad.clone().requires_grad_(True)
        out = ray_helper.ray_marching(None, c, z.clone(), add_inf_z=False, alpha=a, bkg_color=bkg_full)
        loss = (out['rgb'] * g_rgb).sum() + (out['depth'] * g_d).sum() + (out['mask'] * g_m).sum()
        loss.backward()
        tag = 'P{}_alpha'.format(P)
        cases[tag + '_in'] = npy(a)
        for k in ('rgb', 'depth', 'mask', 'trans_shift', 'weights'):
            cases[tag + '_' + k] = npy(out[k])
        cases[tag + '_d_alpha'] = npy(a.grad)
        cases[tag + '_d_radiance'] = npy(c.grad)
    save('g1_compositing', **cases)


# ------------------------------------------------------------------------------------------------
def g2_resampling():
    g = torch.Generator().manual_seed(202)
    R, n_pts, n_sample = 48, 63, 128
    bins = torch.sort(torch.rand(R, n_pts, generator=g) * 4.0 + 0.5, dim=-1)[0]
    weights = torch.rand(R, n_pts - 1, generator=g) ** 4
    weights[::3, 10:40] = 0.0
    # reference cdf (sample_pdf body, ray_helper.py:424-427)
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    det = ray_helper.sample_pdf(bins, weights, n_sample, det=True)
    u_det = torch.linspace(0.0, 1.0, steps=n_sample).expand(R, n_sample).contiguous()
    inds_det = torch.searchsorted(cdf, u_det, right=True)
    # injected u: patch torch.rand for the non-deterministic branch
    u = torch.rand(R, n_sample, generator=g)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: u.clone()
    try:
        rnd = ray_helper.sample_cdf(bins, cdf, n_sample, det=False)
    finally:
        torch.rand = real_rand
    inds_rnd = torch.searchsorted(cdf, u, right=True)
    save('g2_resampling', bins=npy(bins), weights=npy(weights), cdf=npy(cdf), samples_det=npy(det), u_det=npy(u_det),
         inds_det=npy(inds_det).astype(np.int32), u=npy(u), samples_rnd=npy(rnd), inds_rnd=npy(inds_rnd).astype(np.int32))


# ------------------------------------------------------------------------------------------------
def g3_zvals():
    g = torch.Generator().manual_seed(303)
    R = 32
    near = torch.rand(R, 1, generator=g) * 2.0 + 0.1
    far = near + torch.rand(R, 1, generator=g) * 4.0 + 0.01
    out = {'near': npy(near), 'far': npy(far)}
    for n_pts in (2, 64, 129):
        for inclusive in (True, False):
            for inv in (True, False):
                z = ray_helper.get_zvals_from_near_far(near, far, n_pts, inclusive=inclusive, inverse_linear=inv)
                out['n{}_inc{}_inv{}'.format(n_pts, int(inclusive), int(inv))] = npy(z)
    # handle_valid_mask_zvals: compaction of valid samples to the front
    P = 24
    z = torch.sort(torch.rand(R, P, generator=g) * 3 + 1, dim=-1)[0]
    m = torch.rand(R, P, generator=g) > 0.5
    m[0] = False
    m[1] = True
    z[2] = 1.5
    m[2] = True
    z2, m2 = ray_helper.handle_valid_mask_zvals(z.clone(), m.clone())
    out.update(hv_z=npy(z), hv_m=npy(m), hv_z_out=npy(z2), hv_m_out=npy(m2))
    save('g3_zvals', **out)


# ------------------------------------------------------------------------------------------------
def g4_intersections():
    g = torch.Generator().manual_seed(404)
    # hand cases in the spirit of tests/tests_arcnerf/tests_geometry/tests_ray.py:329-349 (side-2 cube at origin)
    o_hand = torch.tensor([[3.0, 0.0, 0.0], [3.0, 0.0, 0.0], [0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.0, 0.0, 0.0],
                           [2.0, 2.0, 0.0], [0.5, 0.5, 3.0], [3.0, 1.0, 1.0]])
    d_hand = torch.tensor([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0],
                           [-1.0, -1.0, 0.0], [0.0, 0.0, -1.0], [-1.0, 0.0, 0.0]])
    d_hand = d_hand / d_hand.norm(dim=-1, keepdim=True)
    n = 1024
    o = torch.randn(n, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * (torch.rand(n, 1, generator=g) * 3.5 + 0.2)
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 3.0
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    rays_o = torch.cat([o_hand, o], 0)
    rays_d = torch.cat([d_hand, d], 0)
    aabb = torch.tensor([[[-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0]], [[-0.5, 1.5], [-1.0, 0.25], [0.0, 2.0]]])  # (2,3,2)
    near, far, pts, mask = aabb_ray_intersection(rays_o, rays_d, aabb, force_torch=True)
    radius = torch.tensor([1.0, 2.5])
    s_near, s_far, s_pts, s_mask = sphere_ray_intersection(rays_o, rays_d, radius=radius)
    save('g4_intersections', rays_o=npy(rays_o), rays_d=npy(rays_d), aabb=npy(aabb), near=npy(near), far=npy(far),
         pts=npy(pts), mask=npy(mask), radius=npy(radius), s_near=npy(s_near), s_far=npy(s_far), s_pts=npy(s_pts),
         s_mask=npy(s_mask))


# ------------------------------------------------------------------------------------------------
def g5_voxel():
    g = torch.Generator().manual_seed(505)
    n = 512
    pts = (torch.rand(n, 3, generator=g) - 0.5) * 2.4  # some outside [-1,1]
    pts[:8] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [0.999999, 0.0, 0.0],
                            [-1.0, 0.5, 0.25], [0.25, 0.25, 0.25], [1.0, 0.0, 0.0], [-1.0000001, 0.0, 0.0]])
    out = {'pts': npy(pts)}
    for n_grid in (8, 15, 128):
        vol = Volume(n_grid=n_grid, side=2.0)
        vidx, valid, gidx, gpts, w = vol.get_voxel_grid_info_from_xyz(pts.clone())
        out['n{}_voxel_idx'.format(n_grid)] = npy(vidx).astype(np.int32)
        out['n{}_valid'.format(n_grid)] = npy(valid)
        out['n{}_corner_idx'.format(n_grid)] = npy(gidx).astype(np.int32)
        out['n{}_weights'.format(n_grid)] = npy(w)
    # occupancy test at n_grid 8 (CPU fallback path, volume.py:929-957)
    vol = Volume(n_grid=8, side=2.0)
    vol.set_up_voxel_bitfield(init_occ=False)
    bf = torch.rand(8, 8, 8, generator=g) > 0.6
    vol.update_bitfield(bf, ops='overwrite')
    occ = vol.check_pts_in_occ_voxel(pts.clone())
    out['n8_bitfield'] = npy(bf)
    out['n8_pts_in_occ'] = npy(occ)
    save('g5_voxel', **out)


# ------------------------------------------------------------------------------------------------
def make_table(n_rows, F, seed, scale):
    """Deterministic table shared with the tests (numpy Generator is platform-stable)."""
    return (np.random.default_rng(seed).random((n_rows, F), dtype=np.float32) * 2.0 - 1.0).astype(np.float32) * np.float32(scale)


def g6_hashgrid():
    g = torch.Generator().manual_seed(606)
    out = {}
    for tag, kw, S in (('ngp', dict(n_levels=16, n_feat_per_entry=2, hashmap_size=19, base_res=16, max_res=2048), 192),
                       ('tiny', dict(n_levels=4, n_feat_per_entry=2, hashmap_size=8, base_res=2, max_res=16), 256),
                       ('f4', dict(n_levels=6, n_feat_per_entry=4, hashmap_size=10, base_res=4, max_res=64), 128)):
        emb = HashGridEmbedder(side=2.0, dtype='torch.float32', include_input=False, backend=None, **kw)
        table = make_table(emb.n_total_embed, emb.n_feat_per_entry, seed=7, scale=0.5)
        emb.embeddings.data = torch.from_numpy(table.copy())
        xyz = (torch.rand(S, 3, generator=g) - 0.5) * 2.2
        xyz[:4] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [0.99999, -0.99999, 0.5]])
        x = xyz.clone().requires_grad_(True)
        y = emb(x)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        dtab = emb.embeddings.grad
        rows = torch.nonzero(dtab.abs().sum(-1) > 0)[:, 0]
        # per-level hash rows recomputed with the reference's own helpers
        idx = np.full((S, len(emb.resolutions), 8), -1, np.int64)
        for i, n_grid in enumerate(emb.resolutions):
            emb.volume.set_n_grid(n_grid, reset_pts=False)
            info = emb.volume.get_voxel_grid_info_from_xyz(xyz.clone())
            valid, gidx = info[1], info[2]
            if gidx is not None:
                h = emb.fast_hash(gidx, emb.offsets[i + 1] - emb.offsets[i]) + emb.offsets[i]
                idx[npy(valid), i, :] = npy(h)
        out.update({
            tag + '_cfg': np.array([kw['n_levels'], kw['n_feat_per_entry'], kw['hashmap_size'], kw['base_res'], kw['max_res']]),
            tag + '_resolutions': np.array(emb.resolutions, np.int32),
            tag + '_offsets': np.array(emb.offsets, np.int64),
            tag + '_xyz': npy(xyz), tag + '_out': npy(y), tag + '_hash_idx': idx.astype(np.int32),
            tag + '_g_out': npy(gy), tag + '_d_xyz': npy(x.grad),
            tag + '_d_table_rows': npy(rows).astype(np.int32), tag + '_d_table_vals': npy(dtab[rows]),
        })
    save('g6_hashgrid', **out)


# ------------------------------------------------------------------------------------------------
def g17_hashgrid_second_order():
    """The input gradient of the encoding differentiated once more (what NeuS on a hash grid needs: normal = d sdf / d x taken
    with create_graph=True, then a loss on the normal).  dx = d<y, gy>/dx ; L2 = <dx, gdx> ; vectors: dL2/d table, dL2/d gy,
    dL2/d x, all from the reference's torch backend by autograd."""
    g = torch.Generator().manual_seed(1717)
    out = {}
    for tag, kw, S in (('ngp', dict(n_levels=16, n_feat_per_entry=2, hashmap_size=19, base_res=16, max_res=2048), 160),
                       ('tiny', dict(n_levels=4, n_feat_per_entry=2, hashmap_size=8, base_res=2, max_res=16), 256),
                       ('f4', dict(n_levels=6, n_feat_per_entry=4, hashmap_size=10, base_res=4, max_res=64), 128)):
        emb = HashGridEmbedder(side=2.0, dtype='torch.float32', include_input=False, backend=None, **kw)
        table = make_table(emb.n_total_embed, emb.n_feat_per_entry, seed=17, scale=0.5)
        emb.embeddings.data = torch.from_numpy(table.copy())
        xyz = (torch.rand(S, 3, generator=g) - 0.5) * 2.2
        xyz[:3] = torch.tensor([[-1.0, -1.0, -1.0], [0.0, 0.0, 0.0], [0.99999, -0.99999, 0.5]])
        x = xyz.clone().requires_grad_(True)
        gy = torch.randn(S, emb.get_output_dim(), generator=g).requires_grad_(True)
        gdx = torch.randn(S, 3, generator=g)
        y = emb(x)
        dx, = torch.autograd.grad((y * gy).sum(), x, create_graph=True)
        l2 = (dx * gdx).sum()
        d_table, d_gy, d_x = torch.autograd.grad(l2, [emb.embeddings, gy, x], allow_unused=True)
        rows = torch.nonzero(d_table.abs().sum(-1) > 0)[:, 0]
        out.update({
            tag + '_cfg': np.array([kw['n_levels'], kw['n_feat_per_entry'], kw['hashmap_size'], kw['base_res'], kw['max_res']]),
            tag + '_xyz': npy(xyz), tag + '_gy': npy(gy), tag + '_gdx': npy(gdx), tag + '_dx': npy(dx),
            tag + '_d_gy': npy(d_gy), tag + '_d_x': npy(d_x if d_x is not None else torch.zeros_like(xyz)),
            tag + '_d_table_rows': npy(rows).astype(np.int32), tag + '_d_table_vals': npy(d_table[rows]),
        })
        print(tag, 'dx', float(dx.abs().max()), 'd_gy', float(d_gy.abs().max()), 'd_x', None if d_x is None else float(d_x.abs().max()),
              'rows', len(rows))
    save('g17_hashgrid_second_order', **out)


# ------------------------------------------------------------------------------------------------
def g20_get_rays():
    """ray generation (render/ray_helper.py:12-119): full image in both flattening orders (with the mip-nerf radius), a pixel
    subset, centre-pixel offset, un-normalised directions, NDC; an intrinsic with skew, a generic pose."""
    from arcnerf.render.ray_helper import get_rays
    g = torch.Generator().manual_seed(2020)
    W, H = 13, 9
    K = torch.tensor([[11.5, 0.3, 6.2], [0.0, 12.25, 4.4], [0.0, 0.0, 1.0]])
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    c2w = torch.eye(4)
    c2w[:3, :3] = q
    c2w[:3, 3] = torch.tensor([0.7, -1.3, 2.9])
    idx = torch.stack([torch.randint(0, W, (17,), generator=g), torch.randint(0, H, (17,), generator=g)], -1)
    out = {'W': np.int32(W), 'H': np.int32(H), 'K': npy(K), 'c2w': npy(c2w), 'index': npy(idx)}
    cases = {'wh': dict(wh_order=True), 'hw': dict(wh_order=False), 'center': dict(wh_order=True, center_pixel=True),
             'raw': dict(wh_order=False, normalize_rays_d=False), 'ndc': dict(wh_order=True, ndc=True, ndc_near=1.0),
             'idx': dict(index=idx), 'idx_center_ndc': dict(index=idx, center_pixel=True, ndc=True, ndc_near=0.5)}
    for tag, kw in cases.items():
        o, d, flat, r = get_rays(W, H, K, c2w, **kw)
        out[tag + '_o'], out[tag + '_d'] = npy(o), npy(d)
        if r is not None:
            out[tag + '_r'] = npy(r)
        if flat is not None:
            out[tag + '_flat'] = np.array(flat, np.int64)
    save('g20_get_rays', **out)


# ------------------------------------------------------------------------------------------------
def g7_freq_sh():
    g = torch.Generator().manual_seed(707)
    S = 128
    x = (torch.rand(S, 3, generator=g) - 0.5) * 4.0
    out = {'x': npy(x)}
    for n_freqs in (10, 4, 0):
        for inc in (True, False):
            if n_freqs == 0 and not inc:
                continue
            xx = x.clone().requires_grad_(True)
            y = FreqEmbedder(3, n_freqs, include_input=inc)(xx)
            gy = torch.randn(y.shape, generator=g)
            (y * gy).sum().backward()
            t = 'freq{}_inc{}'.format(n_freqs, int(inc))
            out[t] = npy(y)
            out[t + '_g'] = npy(gy)
            out[t + '_dx'] = npy(xx.grad)
    d = torch.randn(S, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    out['dirs'] = npy(d)
    for deg in (1, 2, 3, 4, 5):
        for inc in (True, False):
            y = SHEmbedder(n_freqs=deg, include_input=inc, dtype='torch.float32')(d)
            out['sh{}_inc{}'.format(deg, int(inc))] = npy(y)
    save('g7_freq_sh', **out)


# ------------------------------------------------------------------------------------------------
def g8_mlps():
    torch.manual_seed(808)
    g = torch.Generator().manual_seed(809)
    out = {}
    S = 96

    def export(prefix, net):
        for k, v in net.state_dict().items():
            out[prefix + '.' + k] = npy(v)

    # NGP-shaped nets on raw 32-dim features: encoders replaced by identity-like Freq(n_freqs=0)
    for bias in (False, True):
        geo = GeoNet(W=64, D=1, skips=[], W_feat=15, use_bias=bias, geometric_init=False,
                     encoder=dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 32, 'n_freqs': 0}),
                     out_act_cfg=dict_to_obj({'type': 'TruncExp'}))
        x = torch.randn(S, 32, generator=g) * 0.5
        xx = x.clone().requires_grad_(True)
        sig, feat = geo(xx)
        gs, gf = torch.randn(sig.shape, generator=g), torch.randn(feat.shape, generator=g)
        ((sig * gs).sum() + (feat * gf).sum()).backward()
        t = 'geo_b{}'.format(int(bias))
        export(t, geo)
        out.update({t + '_x': npy(x), t + '_sigma': npy(sig), t + '_feat': npy(feat), t + '_g_sigma': npy(gs),
                    t + '_g_feat': npy(gf), t + '_dx': npy(xx.grad)})
        for n, p in geo.named_parameters():
            out[t + '_grad.' + n] = npy(p.grad)

        rad = RadianceNet(mode='fv', W=64, D=2, W_feat_in=16, use_bias=bias,
                          encoder=dict_to_obj({'view': {'type': 'SHEmbedder', 'input_dim': 3, 'n_freqs': 4,
                                                        'include_input': False, 'dtype': 'torch.float32'}}))
        f = torch.randn(S, 16, generator=g)
        v = torch.randn(S, 3, generator=g) * 2.0  # un-normalised on purpose (fuse_radiance_inputs normalises)
        ff = f.clone().requires_grad_(True)
        rgb = rad(None, v, None, ff)
        gr = torch.randn(rgb.shape, generator=g)
        (rgb * gr).sum().backward()
        t = 'rad_b{}'.format(int(bias))
        export(t, rad)
        out.update({t + '_feat': npy(f), t + '_view': npy(v), t + '_rgb': npy(rgb), t + '_g_rgb': npy(gr), t + '_dfeat': npy(ff.grad)})
        for n, p in rad.named_parameters():
            out[t + '_grad.' + n] = npy(p.grad)

    # vanilla-shaped (reduced width) GeoNet with a skip and freq encoder; RadianceNet 'vf' with freq view encoder
    geo = GeoNet(W=32, D=4, skips=[2], W_feat=32, geometric_init=False,
                 encoder=dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 10}))
    x = (torch.rand(S, 3, generator=g) - 0.5) * 3.0
    sig, feat = geo(x)
    export('geo_skip', geo)
    out.update({'geo_skip_x': npy(x), 'geo_skip_sigma': npy(sig), 'geo_skip_feat': npy(feat)})
    rad = RadianceNet(mode='vf', W=16, D=1, W_feat_in=32,
                      encoder=dict_to_obj({'view': {'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 4}}))
    v = torch.randn(S, 3, generator=g)
    rgb = rad(None, v, None, feat.detach())
    export('rad_vf', rad)
    out.update({'rad_vf_view': npy(v), 'rad_vf_rgb': npy(rgb)})

    # TruncExp F1
    te = TruncExp()
    x = torch.linspace(-20.0, 20.0, 81).requires_grad_(True)
    y = te(x)
    y.backward(torch.ones_like(y))
    out.update({'truncexp_x': npy(x), 'truncexp_y': npy(y), 'truncexp_dx': npy(x.grad)})
    save('g8_mlps', **out)


# ------------------------------------------------------------------------------------------------
def g10_occupancy():
    g = torch.Generator().manual_seed(1010)
    vol = Volume(n_grid=8, side=2.0)
    vol.set_up_voxel_bitfield(init_occ=True)
    vol.set_up_voxel_opafield()
    opa0 = torch.rand(8, 8, 8, generator=g) * 0.02
    opa0[torch.rand(8, 8, 8, generator=g) < 0.1] = -1.0  # cells that must never be updated
    vol.opafield = opa0.clone()
    sel = torch.randperm(512, generator=g)[:200]
    vidx = vol.convert_flatten_index_to_xyz_index(sel, 8)
    new = torch.rand(200, generator=g) * 0.03
    vol.update_opafield_by_voxel_idx(vidx, new, ema=0.95)
    opa1 = vol.opafield.clone()
    vol.update_bitfield_by_opafield(threshold=0.01, ops='overwrite')
    save('g10_occupancy', opa0=npy(opa0), flat_idx=npy(sel).astype(np.int32), voxel_idx=npy(vidx).astype(np.int32),
         new_opacity=npy(new), opa1=npy(opa1), bitfield=npy(vol.get_voxel_bitfield()),
         mean_opa=np.float32(vol.get_mean_voxel_opacity()))


def g12_neus():
    """NeuS per-interval math (arcnerf/models/neus_model.py:221-265): sdf_to_cdf / sdf_to_pdf / sdf_to_alpha with the
    gradients w.r.t. the mid sdf, the slope and the (learnable) scale s, both clip settings, duplicated-z tails included."""
    from arcnerf.models.neus_model import sdf_to_alpha, sdf_to_cdf, sdf_to_pdf
    g = torch.Generator().manual_seed(1212)
    R, P = 48, 33
    zvals, _ = torch.sort(torch.rand(R, P, generator=g) * 4.0 + 0.5, dim=-1)
    zvals[::5, -6:] = zvals[::5, -7:-6]                       # padded tails: zero-length intervals
    sdf = (torch.rand(R, P - 1, generator=g) - 0.4) * 0.6
    slope = -torch.rand(R, P - 1, generator=g) * 1.2
    slope[:, ::7] = 0.0
    out = dict(zvals=npy(zvals), mid_sdf=npy(sdf), mid_slope=npy(slope))
    for tag, s_val in (('s64', 64.0), ('s4', 4.0), ('s800', 800.0)):
        out[tag + '_cdf'] = npy(sdf_to_cdf(sdf, s_val))
        out[tag + '_pdf'] = npy(sdf_to_pdf(sdf, s_val))
        for clip in (True, False):
            a = sdf.clone().requires_grad_(True)
            b = slope.clone().requires_grad_(True)
            sv = torch.tensor(s_val, requires_grad=True)
            alpha = sdf_to_alpha(a, zvals, b, sv, clip=clip)
            gout = torch.rand(alpha.shape, generator=torch.Generator().manual_seed(7))
            (alpha * gout).sum().backward()
            key = '{}_clip{}'.format(tag, int(clip))
            out[key + '_alpha'] = npy(alpha)
            out[key + '_d_sdf'], out[key + '_d_slope'], out[key + '_d_s'] = npy(a.grad), npy(b.grad), npy(sv.grad)
            out[key + '_gout'] = npy(gout)
    save('g12_neus', **out)


if __name__ == '__main__':
    if len(sys.argv) > 1:     # regenerate selected vectors only, e.g. `make_golden.py g12_neus`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_compositing()
    g2_resampling()
    g3_zvals()
    g4_intersections()
    g5_voxel()
    g6_hashgrid()
    g7_freq_sh()
    g8_mlps()
    g10_occupancy()
