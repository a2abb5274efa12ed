"""`_volume_func` with the pybind signatures of arcnerf/ops/src/volume_func/volume_func.cpp:277-282: outputs are
caller-allocated tensors written in place, nothing is returned.  RuntimeError on bad inputs like CHECK_INPUT."""
import torch

from arcnerf_amd import _native as N
from arcnerf_amd.ops.volume_func import sampler_rng


def _chk(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')
        if not t.is_contiguous():
            raise RuntimeError('input must be contiguous')


def check_pts_in_occ_voxel(xyz, bitfield, aabb_range, n_grid, output):
    _chk(xyz, bitfield, aabb_range, output)
    N.check(N.lib().arcn_check_pts_in_occ_voxel(xyz.data_ptr(), bitfield.data_ptr(), aabb_range.data_ptr(), int(n_grid),
                                               output.data_ptr(), xyz.shape[0], N.stream()), 'check_pts_in_occ_voxel')


def aabb_intersection(rays_o, rays_d, aabb_range, near, far, pts, mask):
    _chk(rays_o, rays_d, aabb_range, near, far, pts, mask)
    N.check(N.lib().arcn_aabb_intersection(rays_o.data_ptr(), rays_d.data_ptr(), aabb_range.data_ptr(), near.data_ptr(),
                                          far.data_ptr(), pts.data_ptr(), mask.data_ptr(), rays_o.shape[0],
                                          aabb_range.shape[0], N.stream()), 'aabb_intersection')


def sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance, zvals, mask):
    _chk(rays_o, rays_d, near, far, aabb_range, bitfield, zvals, mask)
    rng = sampler_rng()
    N.check(N.lib().arcn_sparse_volume_sampling(rays_o.data_ptr(), rays_d.data_ptr(), near.data_ptr(), far.data_ptr(),
                                               int(n_pts), float(dt), aabb_range.data_ptr(), int(n_grid), bitfield.data_ptr(),
                                               float(near_distance), rng.state, rng.inc, zvals.data_ptr(), mask.data_ptr(),
                                               None, rays_o.shape[0], N.stream()), 'sparse_volume_sampling')
    rng.advance()


def tensor_reduce_max(full_tensor, group_idx, n_group, uni_tensor):
    _chk(full_tensor, group_idx, uni_tensor)
    if group_idx.dtype != torch.int64:
        raise RuntimeError('group_idx must be int64')
    N.check(N.lib().arcn_tensor_reduce_max(full_tensor.data_ptr(), group_idx.data_ptr(), int(n_group), uni_tensor.data_ptr(),
                                          full_tensor.shape[0], N.stream()), 'tensor_reduce_max')
