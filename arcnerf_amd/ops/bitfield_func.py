"""Python wrappers with the reference's names and semantics (arcnerf/ops/bitfield_func.py:16-206), backed by
libarcnerf_hip.so instead of the `_bitfield_func` CUDA extension.  CUDA_BACKEND_AVAILABLE keeps its name: BitfieldBound
asserts on it (obj_bound/bitfield_bound.py:24)."""
import os

import torch

from .. import _native
from . import functional as F

CUDA_BACKEND_AVAILABLE = os.path.exists(_native.LIB_PATH)
HIP_BACKEND_AVAILABLE = CUDA_BACKEND_AVAILABLE

# `_bitfield_func` is its own translation unit in the reference, so it owns ONE file-static `pcg32 rng{9121}`
# (include/common.h:22-23) shared by the sampler and generate_grid_samples and advanced 2^32 after each launch of either
# (bitfield_func_kernel.cu:134, :210).  Explicit and resettable here.
_rng = None


def bitfield_rng(reset=False, seed=9121):
    global _rng
    if _rng is None or reset:
        _rng = F.Pcg32Host(seed)
    return _rng


@torch.no_grad()
def sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance=0.0):
    """aabb_range (3,2) like Volume.get_range(); bitfield (n_grid**3/8) uint8 in Morton order.
    Returns zvals (N_rays, n_pts) (tail = last valid zval) and mask (N_rays, n_pts) bool."""
    rng = bitfield_rng()
    zvals, mask = F.sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, aabb_range.permute(1, 0).contiguous(),
                                               n_grid, bitfield, near_distance, rng.state, rng.inc)
    rng.advance()
    return zvals, mask


@torch.no_grad()
def generate_grid_samples(density_grid, n_elements, density_grid_ema_step, n_grid, thresh):
    """positions (n,3) in [0,1), indices (n,) int32.  An empty request launches nothing and leaves the generator alone
    exactly like the reference (the host `rng.advance()` still runs there: bitfield_func_kernel.cu:205-211)."""
    rng = bitfield_rng()
    pos, idx = F.generate_grid_samples(density_grid, n_elements, density_grid_ema_step, n_grid, thresh, rng.state, rng.inc)
    rng.advance()
    return pos, idx


@torch.no_grad()
def splat_grid_samples(density, density_grid_indices, n_samples, density_grid_tmp):
    return F.splat_grid_samples(density, density_grid_indices, n_samples, density_grid_tmp)


@torch.no_grad()
def ema_grid_samples_nerf(density_grid_tmp, density_grid, n_elements, decay):
    return F.ema_grid_samples_nerf(density_grid_tmp, density_grid, n_elements, decay)


@torch.no_grad()
def update_bitfield(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid):
    return F.update_bitfield(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid)


@torch.no_grad()
def count_bitfield(density_grid_bitfield, n_grid):
    """float count like the reference (CountBitfield.forward returns float(counter[0].item()))"""
    return float(F.count_bitfield(density_grid_bitfield, n_grid)[0].item())
