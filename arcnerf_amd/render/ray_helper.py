"""z sampling helpers and alpha compositing with the reference's function names and return conventions
(arcnerf/render/ray_helper.py:175-620,753-814), backed by the HIP kernels.

Randomness: perturbation draws uniforms through `uniform()` below, one call per reference `torch.rand` call and in the same
order (ray_helper.py:375 perturb_interval, :453 sample_cdf), so a parity run with perturb=True can feed it the reference's recorded
draws (tests/rand_feed.py; golden G23 / G18 / G25); the generator itself cannot be matched bit for bit across devices.
"""
import torch

from ..ops import functional as F
from ..ops.autograd import RayMarchingFn


def get_rays(W, H, intrinsic, c2w, wh_order=True, index=None, n_rays=-1, to_np=False, ndc=False, ndc_near=1.0, center_pixel=False,
             normalize_rays_d=True):
    """Rays of one camera in world coordinates, one kernel (render/ray_helper.py:12-119).
    intrinsic (3,3), c2w (4,4) device tensors.  index: (N,2) (i, j) pixel pairs; n_rays > 0: that many distinct random pixels
    (numpy's global generator, like the reference); neither: the full image, flattened column-major (wh_order) or row-major.
    -> rays_o (N,3), rays_d (N,3), index (list of column-major pixel ids, or None), rays_r (N,1) mip-nerf radius (full image only)"""
    assert (index is None) or n_rays <= 0, 'You are not allowed to sampled both by index and N_ray'
    device = intrinsic.device
    flat = None
    if index is not None:
        assert len(index.shape) == 2 and index.shape[-1] == 2, 'invalid shape, should be (N_rays, 2)'
        index = torch.as_tensor(index, dtype=torch.long, device=device)
        flat = index[:, 0] * H + index[:, 1]
    if n_rays > 0:
        import numpy as np
        flat = torch.tensor(np.random.choice(range(0, W * H), n_rays, replace=False), dtype=torch.long, device=device)
    rays_o, rays_d, rays_r = F.get_rays(W, H, intrinsic, c2w, wh_order=wh_order, flat_index=flat, center_pixel=center_pixel,
                                        normalize_rays_d=normalize_rays_d, ndc=ndc, ndc_near=ndc_near)
    if to_np:
        rays_o, rays_d = rays_o.cpu().numpy(), rays_d.cpu().numpy()
    return rays_o, rays_d, (flat.cpu().numpy().tolist() if flat is not None else None), rays_r


def get_near_far_from_rays(rays_o, rays_d, bounds=None, near_hardcode=None, far_hardcode=None, bounding_radius=None):
    """near, far (N_rays, 1) from data bounds and / or the ray-sphere test of a bounding radius, overridden by the hard-coded
    values (ray_helper.py:175-228)"""
    n_rays = rays_o.shape[0]
    if near_hardcode is None or far_hardcode is None:
        if bounds is None and bounding_radius is None:
            raise NotImplementedError('You must specify near/far in some place...')
        if bounds is None:
            from ..geometry.ray import sphere_ray_intersection
            near, far, _, _ = sphere_ray_intersection(rays_o, rays_d, float(bounding_radius))
        else:
            near, far = bounds[:, 0:1], bounds[:, 1:2]
            if bounding_radius is not None:  # the sphere restricts the far end
                from ..geometry.ray import sphere_ray_intersection
                far_bound = sphere_ray_intersection(rays_o, rays_d, float(bounding_radius))[1]
                far = torch.where(far > far_bound, far_bound, far)
        if near_hardcode is not None:
            near = near * 0 + near_hardcode
        if far_hardcode is not None:
            far = far * 0 + far_hardcode
    else:
        near = torch.full((n_rays, 1), float(near_hardcode), dtype=rays_o.dtype, device=rays_o.device)
        far = torch.full((n_rays, 1), float(far_hardcode), dtype=rays_o.dtype, device=rays_o.device)
    far = torch.where(far <= near, near + 1e-5, far)
    return near, far


def uniform(shape, dtype, device):
    """U[0, 1) draws of the sampling helpers: the ONE place they come from (see the module docstring)"""
    return torch.rand(tuple(shape), dtype=dtype, device=device)


def perturb_interval(vals):
    mids = 0.5 * (vals[..., 1:] + vals[..., :-1])
    upper = torch.cat([mids, vals[..., -1:]], -1)
    lower = torch.cat([vals[..., :1], mids], -1)
    return lower + (upper - lower) * uniform(upper.shape, upper.dtype, upper.device)


def get_zvals_from_near_far(near, far, n_pts, inclusive=True, inverse_linear=False, perturb=False):
    if inclusive:
        t = torch.linspace(0.0, 1.0, n_pts, dtype=near.dtype, device=near.device)
    else:
        t = torch.linspace(0.0, 1.0, n_pts + 2, dtype=near.dtype, device=near.device)[1:-1]
    if inverse_linear:
        zvals = 1.0 / (1.0 / (near + 1e-8) * (1.0 - t) + 1.0 / (far + 1e-8) * t)
    else:
        zvals = near + (far - near) * t
    return perturb_interval(zvals) if perturb else zvals


def get_zvals_from_sphere_radius(rays_o, rays_d, sphere_radius):
    """far intersection of every ray with every sphere shell (ray_helper.py:343-358); 0 where a ray misses a shell"""
    from ..geometry.ray import sphere_ray_intersection
    return sphere_ray_intersection(rays_o, rays_d, sphere_radius)[1]


def get_zvals_outside_sphere(rays_o, rays_d, n_pts, radius, perturb=False):
    """multi-sphere sampling outside a bounding radius, shells at radius / t for t in (0,1) (ray_helper.py:318-340):
    -> zvals (N_rays, N_pts), sphere_radius (N_pts,)"""
    t_vals = torch.linspace(0.0, 1.0, n_pts + 2, dtype=rays_o.dtype, device=rays_o.device)[1:-1]
    sphere_radius = radius / torch.flip(t_vals, dims=[-1])
    if perturb:
        sphere_radius = perturb_interval(sphere_radius[None])[0]
    return get_zvals_from_sphere_radius(rays_o, rays_d, sphere_radius), sphere_radius


def perturb_interval_with_mask(vals, mask=None):
    """perturb only the valid prefix of each row; the masked tail keeps following the last valid value
    (ray_helper.py:378-407)"""
    pert = perturb_interval(vals)
    if mask is None:
        return pert
    vals = torch.where(mask, pert, vals)
    last = (mask.sum(dim=1) - 1).clamp(min=0)
    last_val = vals.gather(1, last[:, None])
    return torch.minimum(torch.maximum(vals, vals[:, 0:1]), last_val)


def get_zvals_from_near_far_fix_step(near, far, fix_t, n_pts, inclusive=True, perturb=False):
    """zvals = near + k*fix_t clamped to far, duplicate tail masked out (ray_helper.py:267-315).  This is the reference's
    torch fallback when its CUDA sampler is missing; kept for API completeness — VolumeBound always uses the marcher here.
    NB the reference perturbs unconditionally (`if perturb or True`, :312); `perturb=False` gives the deterministic grid."""
    assert fix_t > 0, 'Only allow positive step...'
    step = torch.arange(n_pts, device=near.device, dtype=near.dtype)[None]
    zvals = (near if inclusive else near + fix_t) + step * fix_t
    zvals = torch.minimum(torch.maximum(zvals, near), far)
    same = torch.cat([torch.zeros_like(zvals[:, :1], dtype=torch.bool), (zvals[:, 1:] - zvals[:, :-1]) == 0.0], dim=1)
    mask_pts = ~same
    if perturb:
        zvals = perturb_interval_with_mask(zvals, mask_pts)
    return zvals, mask_pts


def handle_valid_mask_zvals(zvals, mask):
    """Move the valid samples of every ray to the front (stable), pad the tail with the last valid z, rows without any
    valid sample become all-zero, constant fully-valid rows keep one sample (ray_helper.py:753-814).  Sort-free: the
    destination column of a valid sample is the running count of valid samples before it."""
    assert zvals.dim() == 2 and zvals.shape == mask.shape, 'Both tensor should be in (B, N)'
    zvals, mask = zvals.clone(), mask.clone()
    none = ~mask.any(dim=1)
    zvals[none] = 0.0
    const = (torch.abs(zvals[:, 1:] - zvals[:, :-1]) < 1e-7).all(dim=1) & mask.all(dim=1)
    mask[const, 1:] = False
    count = mask.sum(dim=1)
    dest = torch.cumsum(mask.long(), dim=1) - 1
    P = zvals.shape[1]
    packed = torch.zeros_like(zvals)
    packed.scatter_(1, torch.where(mask, dest, torch.full_like(dest, P - 1)), torch.where(mask, zvals, torch.zeros_like(zvals)))
    # the scatter above may have dropped zeros on column P-1 of rows that are not full: rebuild from the last valid value
    last_val = packed.gather(1, (count - 1).clamp(min=0)[:, None])
    cols = torch.arange(P, device=zvals.device)[None]
    out_mask = cols < count[:, None]
    full_last = torch.where(count[:, None] == P, zvals.gather(1, torch.full_like(count, P - 1)[:, None]), last_val)
    # column P-1 of a full row holds its own (last) sample; every other row pads with its last valid value
    packed = torch.where(out_mask, packed, last_val.expand_as(packed))
    packed[:, P - 1:] = torch.where(count[:, None] == P, full_last, packed[:, P - 1:])
    packed[none] = 0.0
    keep = ~(none | const)
    zvals = torch.where(keep[:, None], packed, zvals)
    mask = torch.where(keep[:, None], out_mask, mask)
    return zvals, mask


def sample_cdf(bins, cdf, n_sample, det=False, eps=1e-5):
    """inverse-CDF sampling + per-row sort, one kernel (searchsorted right=True semantics)"""
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_sample, device=bins.device).expand(cdf.shape[0], n_sample)
    else:
        u = uniform((cdf.shape[0], n_sample), torch.float32, bins.device)
    return F.sample_cdf(bins, cdf.detach(), u.contiguous(), eps=eps, sort=True)


_LATTICE = {}


def sample_pdf(bins, weights, n_sample, det=False, eps=1e-5):
    """(ray_helper.py:410-429) weights -> pdf -> cdf -> inverse CDF -> sorted samples: ONE kernel (arcn_sample_pdf; the cdf accumulated
    in double and rounded per prefix like torch's CPU cumsum) instead of add / sum / div / cumsum / cat / searchsorted / gathers / sort"""
    if det:
        key = (int(n_sample), bins.device)
        if key not in _LATTICE:
            _LATTICE[key] = torch.linspace(0.0, 1.0, steps=n_sample, device=bins.device)[None].contiguous()
        u = _LATTICE[key]
    else:
        u = uniform((bins.shape[0], n_sample), torch.float32, bins.device).contiguous()
    return F.sample_pdf(bins.detach().contiguous().float(), weights.detach().contiguous().float(), u, eps=eps, sort=True)


def alpha_to_weights(alpha):
    """trans_shift, weights (N_rays, N_p) from alpha — the same compositing kernel with sigma := alpha branch"""
    out = F.ray_marching_fwd(None, None, torch.zeros_like(alpha), alpha=alpha, add_inf_z=False)
    return out['trans_shift'], out['weights']


def ray_marching(sigma, radiance, zvals, add_inf_z=False, noise_std=0.0, weights_only=False, white_bkg=False, alpha=None,
                 bkg_color=None):
    """Dict with rgb (N_rays,3) | None, depth, mask (N_rays), and per-sample sigma/radiance/zvals/alpha/trans_shift/weights
    (N_rays, N_pts or N_pts-1) exactly as the reference returns them.  Differentiable wrt sigma|alpha and radiance."""
    assert sigma is not None or alpha is not None, 'Can not be None for both alpha and sigma..'
    R, P = zvals.shape
    drop_last = (not add_inf_z) and alpha is None
    noise = None
    if alpha is None and noise_std > 0.0:
        Pe = P - 1 if drop_last else P
        noise = torch.randn((R, Pe), dtype=zvals.dtype, device=zvals.device) * noise_std
    if bkg_color is not None:
        assert bkg_color.shape[0] == R or bkg_color.shape[0] == 1, 'Only bkg with N_rays/1 allowed..'
    rgb, depth, mask, a, trans, w, status, t_last = RayMarchingFn.apply(sigma, radiance, zvals.contiguous(), alpha, bkg_color,
                                                                       noise, bool(add_inf_z), bool(white_bkg))
    if torch.is_grad_enabled() and t_last.requires_grad:
        # the last column carries gradient: FullModel.blend_bkg_rgb scales the background model with trans_shift[:, -1]
        trans = torch.cat([trans[:, :-1], t_last[:, None]], dim=1)
    _LAST_STATUS['t'] = status
    if weights_only:
        return {'weights': w}
    _sigma = sigma[:, :-1] if (drop_last and sigma is not None) else sigma
    _radiance = radiance[:, :-1, :] if (drop_last and radiance is not None) else radiance
    _zvals = zvals[:, :-1] if drop_last else zvals
    return {'rgb': rgb if radiance is not None else None, 'depth': depth, 'mask': mask, 'sigma': _sigma, 'radiance': _radiance,
            'zvals': _zvals, 'alpha': a, 'trans_shift': trans, 'weights': w}


_LAST_STATUS = {'t': None}


def last_order_status():
    """Device int32 flag of the most recent ray_marching call: 1 if some z decreased along a ray.  The reference asserts
    this on the host (`assert torch.all(deltas >= 0)`, ray_helper.py:534) which costs a device sync per call; here the
    check is a by-product of the kernel and is read only when somebody asks."""
    return _LAST_STATUS['t']
