"""Python wrappers with the reference's names and semantics (arcnerf/ops/multivol_func.py:16-141), backed by
libarcnerf_hip.so instead of the `_multivol_func` CUDA extension."""
import os

import torch

from .. import _native
from . import functional as F

CUDA_BACKEND_AVAILABLE = os.path.exists(_native.LIB_PATH)
HIP_BACKEND_AVAILABLE = CUDA_BACKEND_AVAILABLE

# `_multivol_func` is its own translation unit: one file-static `pcg32 rng{9121}` (include/common.h:22-23) shared by the
# sampler and generate_grid_samples_multivol, advanced 2^32 after each launch (multivol_func_kernel.cu:144, :238).
_rng = None


def multivol_rng(reset=False, seed=9121):
    global _rng
    if _rng is None or reset:
        _rng = F.Pcg32Host(seed)
    return _rng


@torch.no_grad()
def sparse_sampling_in_multivol_bitfield(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb_range,
                                         aabb_range, n_grid, n_cascade, bitfield, near_distance=0.0, inclusive=False):
    """min_aabb_range / aabb_range (3,2) like Volume.get_range(): inner and outermost volume.
    Returns zvals (N_rays, n_pts) (tail = last valid zval) and mask (N_rays, n_pts) bool."""
    rng = multivol_rng()
    zvals, mask = F.sparse_sampling_in_multivol_bitfield(
        rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb_range.permute(1, 0).contiguous(),
        aabb_range.permute(1, 0).contiguous(), n_grid, n_cascade, bitfield, near_distance, inclusive, rng.state, rng.inc)
    rng.advance()
    return zvals, mask


@torch.no_grad()
def generate_grid_samples_multivol(density_grid, n_elements, aabb_range, density_grid_ema_step, n_cascade, n_grid, thresh,
                                   inclusive):
    """aabb_range (3,2): inner volume.  positions (n,3) in world space, indices (n,) int32."""
    rng = multivol_rng()
    pos, idx = F.generate_grid_samples_multivol(density_grid, n_elements, aabb_range.permute(1, 0).contiguous(),
                                                density_grid_ema_step, n_cascade, n_grid, thresh, inclusive, rng.state, rng.inc)
    rng.advance()
    return pos, idx


@torch.no_grad()
def update_bitfield_multivol(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid, n_cascade, inclusive):
    return F.update_bitfield_multivol(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid, n_cascade, inclusive)
