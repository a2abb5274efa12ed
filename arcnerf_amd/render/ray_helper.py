"""z sampling helpers and alpha compositing with the reference's function names and return conventions
(arcnerf/render/ray_helper.py:175-620,753-814), backed by the HIP kernels.

Randomness: perturbation uses torch.rand like the reference (cannot be matched bit-for-bit, parity runs use
perturb=False / inference_only=True).
"""
import torch

from ..ops import functional as F
from ..ops.autograd import RayMarchingFn


def get_near_far_from_rays(rays_o, rays_d, bounds=None, near_hardcode=None, far_hardcode=None, bounding_radius=None):
    """near, far (N_rays, 1).  Sphere bounding (bounding_radius) belongs to the NeuS row and is not provided yet."""
    n_rays = rays_o.shape[0]
    if near_hardcode is None or far_hardcode is None:
        if bounds is None and bounding_radius is None:
            raise NotImplementedError('You must specify near/far in some place...')
        if bounds is None or bounding_radius is not None:
            raise NotImplementedError('ray-sphere bounds are part of the NeuS row (next), not of this path yet')
        near, far = bounds[:, 0:1], bounds[:, 1:2]
        if near_hardcode is not None:
            near = near * 0 + near_hardcode
        if far_hardcode is not None:
            far = far * 0 + far_hardcode
    else:
        near = torch.full((n_rays, 1), float(near_hardcode), dtype=rays_o.dtype, device=rays_o.device)
        far = torch.full((n_rays, 1), float(far_hardcode), dtype=rays_o.dtype, device=rays_o.device)
    far = torch.where(far <= near, near + 1e-5, far)
    return near, far


def perturb_interval(vals):
    mids = 0.5 * (vals[..., 1:] + vals[..., :-1])
    upper = torch.cat([mids, vals[..., -1:]], -1)
    lower = torch.cat([vals[..., :1], mids], -1)
    return lower + (upper - lower) * torch.rand_like(upper)


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


def sample_cdf(bins, cdf, n_sample, det=False, eps=1e-5):
    """inverse-CDF sampling + per-row sort, one kernel (searchsorted right=True semantics)"""
    if det:
        u = torch.linspace(0.0, 1.0, steps=n_sample, device=bins.device).expand(cdf.shape[0], n_sample)
    else:
        u = torch.rand(cdf.shape[0], n_sample, device=bins.device)
    return F.sample_cdf(bins, cdf.detach(), u.contiguous(), eps=eps, sort=True)


def sample_pdf(bins, weights, n_sample, det=False, eps=1e-5):
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    return sample_cdf(bins, cdf, n_sample, det, eps)


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
    rgb, depth, mask, a, trans, w, status = RayMarchingFn.apply(sigma, radiance, zvals.contiguous(), alpha, bkg_color, noise,
                                                               bool(add_inf_z), bool(white_bkg))
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
