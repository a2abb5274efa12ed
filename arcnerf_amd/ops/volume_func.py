"""Python wrappers with the reference's names and semantics (arcnerf/ops/volume_func.py:15-167), backed by
libarcnerf_hip.so instead of the `_volume_func` CUDA extension.  CUDA_BACKEND_AVAILABLE keeps its name: it is what
geometry/ray.py and obj_bound/volume_bound.py test to select the native path."""
import os

import torch

from .. import _native
from . import functional as F

CUDA_BACKEND_AVAILABLE = os.path.exists(_native.LIB_PATH)
HIP_BACKEND_AVAILABLE = CUDA_BACKEND_AVAILABLE

# the reference keeps one process-global `static pcg32 rng{9121}` advanced 2^32 after every sampling launch
# (include/common.h:22-23, volume_func_kernel.cu:283-289); same default here, but explicit and resettable.
_rng = None


def sampler_rng(reset=False, seed=9121):
    global _rng
    if _rng is None or reset:
        _rng = F.Pcg32Host(seed)
    return _rng


@torch.no_grad()
def check_pts_in_occ_voxel_cuda(xyz, bitfield, aabb_range, n_grid):
    """aabb_range (3,2) min/max like Volume.get_range()"""
    return F.check_pts_in_occ_voxel(xyz, bitfield, aabb_range.permute(1, 0).contiguous(), n_grid)


@torch.no_grad()
def ray_aabb_intersection_cuda(rays_o, rays_d, aabb_range):
    """aabb_range (N_v,3,2) -> near, far (N_rays,N_v), pts (N_rays,N_v,2,3), mask (N_rays,N_v)  [K2: mask = tmin > 0]"""
    return F.aabb_intersection(rays_o, rays_d, aabb_range.permute(0, 2, 1).contiguous())


@torch.no_grad()
def sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance=0.0):
    rng = sampler_rng()
    zvals, mask = F.sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, aabb_range.permute(1, 0).contiguous(),
                                           n_grid, bitfield, near_distance, rng.state, rng.inc)
    rng.advance()
    return zvals, mask


@torch.no_grad()
def tensor_reduce_max(full_tensor, idx, n_group):
    return F.tensor_reduce_max(full_tensor, idx, n_group)
