"""Ray / implicit-surface intersection for surface rendering (arcnerf/geometry/ray.py:350-601): sphere tracing for sdf nets,
bracket + secant refinement for any level set.  Inference only.

Device-resident formulation: the reference gathers the still-active rays with boolean masks every iteration (two host
round-trips per step); here every iteration evaluates the field on ALL rays at their current point and applies the update
under the masks.  A frozen ray (converged, or out of [near, far]) keeps its point, so re-evaluating it changes nothing and the
final zvals / masks are the same; the only host reads are the early-exit checks (every few iterations for sphere tracing, the
reference's own per-iteration criterion for the secant loop, which decides how many refinements run)."""
import torch

from .ray import get_ray_points_by_zvals


def _per_ray(value, n_rays, like):
    """near / far as (n_rays, 1): a tensor of that shape is used as is, anything else is broadcast"""
    if torch.is_tensor(value) and tuple(value.shape) == (n_rays, 1):
        return value
    return torch.ones((n_rays, 1), dtype=like.dtype, device=like.device) * value


@torch.no_grad()
def sphere_tracing(rays_o, rays_d, sdf_func, near=0.0, far=10.0, n_iter=100, threshold=0.001, check_every=4):
    """March t += sdf(o + t d) from `near` until |sdf| < threshold.
    -> zvals (N,1) (0 where <= near), pts (N,3), mask (N,) bool: the ray never left [near, far] (ray.py:403-467)"""
    n_rays = rays_o.shape[0]
    near_t, far_t = _per_ray(near, n_rays, rays_o), _per_ray(far, n_rays, rays_o)
    zvals = near_t.clone()
    in_range = torch.ones(n_rays, dtype=torch.bool, device=rays_o.device)
    on_surface = torch.zeros_like(in_range)
    for it in range(n_iter):
        sdf = sdf_func(get_ray_points_by_zvals(rays_o, rays_d, zvals).view(-1, 3))
        on_surface = on_surface | (sdf.abs() < threshold)
        marching = in_range & ~on_surface
        zvals = torch.where(marching[:, None], zvals + sdf[:, None], zvals)
        in_range = in_range & ~(zvals[:, 0] > far_t[:, 0]) & ~(zvals[:, 0] < near_t[:, 0])
        if it % check_every == check_every - 1 and not bool((in_range & ~on_surface).any()):
            break
    zvals = torch.where(zvals <= near_t, torch.zeros_like(zvals), zvals)
    return zvals, get_ray_points_by_zvals(rays_o, rays_d, zvals).view(-1, 3), in_range


@torch.no_grad()
def secant_root_finding(rays_o, rays_d, geo_func, near=0.0, far=10.0, n_step=128, n_iter=20, threshold=0.001, level=0.0,
                        grad_dir='ascent'):
    """First crossing of `level` from outside to inside among n_step uniform samples, refined by the secant rule.
    grad_dir 'ascent': inside is below the level (sdf); 'descent': inside is above it (density).
    -> zvals (N,1): the root; `far` without a crossing; 0 when the ray starts inside or the root is <= near; pts; mask
    (ray.py:470-601)"""
    n_rays, dev = rays_o.shape[0], rays_o.device
    near_t, far_t = _per_ray(near, n_rays, rays_o), _per_ray(far, n_rays, rays_o)
    frac = torch.linspace(0., 1., n_step, device=dev)[None, :]
    t_grid = near_t * (1 - frac) + far_t * frac
    sign = -1.0 if grad_dir == 'descent' else 1.0    # after the flip: positive outside, negative inside

    def field(z):
        return (geo_func(get_ray_points_by_zvals(rays_o, rays_d, z).view(-1, 3)).view(n_rays, -1) - level) * sign

    g = field(t_grid)
    starts_outside = g[:, 0] > 0
    # the first interval whose ends differ in sign: sign products weighted n_step .. 1, so the earliest change is the minimum
    flips = torch.cat([torch.sign(g[:, :-1] * g[:, 1:]), torch.ones((n_rays, 1), device=dev)], dim=-1)
    cost, first = torch.min(flips * torch.arange(n_step, 0, -1, dtype=rays_o.dtype, device=dev), dim=-1)
    row = torch.arange(n_rays, device=dev)
    nxt = torch.clamp(first + 1, max=n_step - 1)
    mask = starts_outside & (cost < 0) & (g[row, first] > 0)
    z_out, g_out = t_grid[row, first], g[row, first]      # still outside (positive side)
    z_in, g_in = t_grid[row, nxt], g[row, nxt]            # already inside (negative side)

    def secant():
        return -g_in * (z_out - z_in) / (g_out - g_in) + z_in

    if bool(mask.any()):
        z_mid = secant()
        z_first = z_mid.clone()
        for i in range(n_iter):
            # the reference's stopping rule compares against the FIRST estimate (ray.py:566-568): kept, it sets the iteration count
            if i > 0 and bool(torch.all(torch.where(mask, (z_first - z_mid).abs(), torch.zeros_like(z_mid)) < threshold)):
                break
            g_mid = field(z_mid[:, None])[:, 0]
            inside = g_mid < 0
            z_in, g_in = torch.where(inside, z_mid, z_in), torch.where(inside, g_mid, g_in)
            # The outside end never moves: the reference guards that update with `~ind_low.sum() > 0` (ray.py:583), the bitwise
            # NOT of a count, which is never positive.  An estimate that lands outside therefore stalls that ray; reproduced,
            # since the results have to be the reference's.
            z_mid = secant()
        zvals = torch.where(mask, z_mid, far_t[:, 0])[:, None]
    else:
        zvals = far_t.clone()
    zvals = torch.where(starts_outside[:, None], zvals, torch.zeros_like(zvals))
    zvals = torch.where(zvals <= near_t, torch.zeros_like(zvals), zvals)
    return zvals, get_ray_points_by_zvals(rays_o, rays_d, zvals).view(-1, 3), mask


def surface_ray_intersection(rays_o, rays_d, geo_func, method='sphere_tracing', near=0.0, far=10.0, n_step=128, n_iter=100,
                             threshold=0.001, level=0.0, grad_dir='ascent'):
    """-> zvals (N,1), pts (N,3), mask (N,) bool (ray.py:350-398)"""
    if method == 'sphere_tracing':
        return sphere_tracing(rays_o, rays_d, geo_func, near, far, n_iter, threshold)
    if method == 'secant_root_finding':
        return secant_root_finding(rays_o, rays_d, geo_func, near, far, n_step, n_iter, threshold, level, grad_dir)
    raise NotImplementedError('Method {} not support for surface-ray intersection'.format(method))
