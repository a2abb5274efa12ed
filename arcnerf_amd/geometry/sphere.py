"""Sphere (arcnerf/geometry/sphere.py:268-337): the parts of the class that sit on the rendering path - origin / radius
parameters and the ray test.  Mesh / line / point generation helpers of the reference class are visualisation code
(SURVEY.md section 2.1 OUT)."""
import torch
import torch.nn as nn

from .ray import sphere_ray_intersection


class Sphere(nn.Module):
    def __init__(self, origin=(0, 0, 0), radius=1.0, dtype=torch.float32, requires_grad=False):
        super().__init__()
        self.dtype = dtype
        self.requires_grad = requires_grad
        self.origin = nn.Parameter(torch.tensor([0.0, 0.0, 0.0], dtype=dtype), requires_grad=requires_grad)
        self.radius = nn.Parameter(torch.tensor([0.0], dtype=dtype), requires_grad=requires_grad)
        self.set_params(origin, radius)

    @torch.no_grad()
    def set_params(self, origin, radius):
        self.set_origin(origin)
        self.set_radius(radius)

    @torch.no_grad()
    def set_origin(self, origin=(0.0, 0.0, 0.0)):
        for k in range(3):
            self.origin[k] = origin[k]

    def _host_value(self, name, src, make):
        """Python numbers of a small device tensor, read back once per version of it (every read is a host / device synchronisation and
        the ray-sphere intersection asks on every call); the in-place setters move the key."""
        key = (src.data_ptr(), src._version, str(src.device))
        if src.requires_grad:      # a learnable tensor is updated by raw-pointer kernels (FusedAdam) that do not move its version: no cache
            return make()
        cache = self.__dict__.setdefault('_host_cache', {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = cache[name] = (key, make())
        return hit[1]

    def get_origin(self, in_tuple=False):
        """origin as a (3,) tensor, or a tuple of floats"""
        if in_tuple:
            return self._host_value('origin', self.origin, lambda: tuple(float(v) for v in self.origin.detach().cpu().tolist()))
        return self.origin

    @torch.no_grad()
    def set_radius(self, radius):
        self.radius[0] = radius

    def get_radius(self, in_float=False):
        """radius as a (1,) tensor, or a float"""
        if in_float:
            return self._host_value('radius', self.radius, lambda: float(self.radius.detach().cpu()[0]))
        return self.radius

    def ray_sphere_intersection(self, rays_o, rays_d):
        """-> near, far (N_rays, 1), pts (N_rays, 1, 2, 3), mask (N_rays, 1) bool"""
        return sphere_ray_intersection(rays_o, rays_d, self.get_radius(in_float=True), self.get_origin(in_tuple=True))
