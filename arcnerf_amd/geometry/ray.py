"""Ray helpers of the path: o + t*d point generation, the ray/AABB test and the ray/sphere test.

Mirrors arcnerf/geometry/ray.py:11-30 (get_ray_points_by_zvals), :180-255 (sphere_ray_intersection) and :258-350
(aabb_ray_intersection).  The reference
switches between a CUDA kernel (K2: rays starting inside the box are masked out) and a torch implementation (eps-shifted,
inside rays hit) depending on availability; both semantics are kernels here, `force_torch` selects the torch one.
"""
import torch

from ..ops import functional as F


def get_ray_points_by_zvals(rays_o, rays_d, zvals):
    """pts (N_rays, N_pts, 3) = rays_o + zvals * rays_d"""
    return rays_o.unsqueeze(1) + zvals.unsqueeze(-1) * rays_d.unsqueeze(1)


def normalize(vec):
    """v / (|v| + 1e-8)  (arcnerf/geometry/transformation.py:11-25)"""
    return vec / (torch.norm(vec, dim=-1).unsqueeze(-1) + 1e-8)


@torch.no_grad()
def aabb_ray_intersection(rays_o, rays_d, aabb_range, eps=1e-7, force_torch=False, want_pts=True):
    """aabb_range (N_v, 3, 2) -> near, far (N_rays, N_v), pts (N_rays, N_v, 2, 3) (None without want_pts), mask (N_rays, N_v) bool"""
    assert aabb_range.shape[1] == 3 and aabb_range.shape[2] == 2, 'AABB range must be (N, 3, 2)'
    if force_torch:
        return F.aabb_intersection_torch(rays_o, rays_d, aabb_range, eps, want_pts=want_pts)
    return F.aabb_intersection(rays_o, rays_d, aabb_range.permute(0, 2, 1).contiguous(), want_pts=want_pts)


@torch.no_grad()
def sphere_ray_intersection(rays_o, rays_d, radius, origin=(0, 0, 0)):
    """radius float or (N_r,) -> near, far (N_rays, N_r), pts (N_rays, N_r, 2, 3), mask (N_rays, N_r) bool
    (outside/no hit: 0, 0, mask 0; inside: near 0; rays_d assumed normalised, like the reference)"""
    return F.sphere_intersection(rays_o, rays_d, radius, origin)


def __getattr__(name):
    """surface_ray_intersection / sphere_tracing / secant_root_finding live in geometry/surface.py (which imports this module);
    they stay reachable under the reference's path arcnerf.geometry.ray"""
    if name in ('surface_ray_intersection', 'sphere_tracing', 'secant_root_finding'):
        from . import surface
        return getattr(surface, name)
    raise AttributeError(name)
