"""`_multivol_func` with the pybind signatures of arcnerf/ops/src/multivol_func/multivol_func.cpp: outputs are
caller-allocated tensors written in place, nothing is returned.  RuntimeError on bad inputs like CHECK_INPUT."""
import torch

from arcnerf_amd import _native as N
from arcnerf_amd.ops.multivol_func import multivol_rng


def _chk(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')
        if not t.is_contiguous():
            raise RuntimeError('input must be contiguous')


def _is(t, dtype, what):
    if t.dtype != dtype:
        raise RuntimeError('{} must be {}'.format(what, dtype))


def _levels(n_cascade, inclusive):
    return int(n_cascade) if inclusive else int(n_cascade) - 1


def sparse_sampling_in_multivol_bitfield(rays_o, rays_d, near, far, n_pts, cone_angle, min_step, max_step, min_aabb_range,
                                         aabb_range, n_grid, n_cascade, bitfield, near_distance, inclusive, zvals, mask):
    _chk(rays_o, rays_d, near, far, min_aabb_range, aabb_range, bitfield, zvals, mask)
    for t, w in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (near, 'near'), (far, 'far'), (min_aabb_range, 'min_aabb_range'),
                 (aabb_range, 'aabb_range'), (zvals, 'zvals')):
        _is(t, torch.float32, w)
    _is(bitfield, torch.uint8, 'bitfield')
    _is(mask, torch.bool, 'mask')
    if rays_o.shape[1] != 3 or rays_d.shape[1] != 3:
        raise RuntimeError('Input rays tensor must be (B, 3).')
    if near.shape[1] != 1 or far.shape[1] != 1:
        raise RuntimeError('Input near/far tensor must be (B, 1).')
    if tuple(aabb_range.shape) != (2, 3) or tuple(min_aabb_range.shape) != (2, 3):
        raise RuntimeError('xyz range should be in (2, 3).')
    if bitfield.shape[0] != n_grid * n_grid * n_grid // 8 * _levels(n_cascade, inclusive):
        raise RuntimeError('bitfield should be in (n_grid**3/8 * levels,).')
    if tuple(zvals.shape) != (rays_o.shape[0], n_pts) or tuple(mask.shape) != (rays_o.shape[0], n_pts):
        raise RuntimeError('zval / mask should be in (n_rays, n_pts).')
    rng = multivol_rng()
    N.check(N.lib().arcn_sparse_sampling_in_multivol_bitfield(
        rays_o.data_ptr(), rays_d.data_ptr(), near.data_ptr(), far.data_ptr(), int(n_pts), float(cone_angle), float(min_step),
        float(max_step), min_aabb_range.data_ptr(), aabb_range.data_ptr(), int(n_grid), int(n_cascade), bitfield.data_ptr(),
        float(near_distance), int(bool(inclusive)), rng.state, rng.inc, zvals.data_ptr(), mask.data_ptr(), None, rays_o.shape[0],
        N.stream()), 'sparse_sampling_in_multivol_bitfield')
    rng.advance()


def generate_grid_samples_multivol(density_grid, density_grid_ema_step, n_elements, aabb_range, n_cascade, n_grid, thresh,
                                   inclusive, density_grid_positions_uniform, density_grid_indices_uniform):
    _chk(density_grid, aabb_range, density_grid_positions_uniform, density_grid_indices_uniform)
    _is(density_grid, torch.float32, 'density_grid')
    _is(aabb_range, torch.float32, 'aabb_range')
    _is(density_grid_positions_uniform, torch.float32, 'positions')
    _is(density_grid_indices_uniform, torch.int32, 'indices')
    rng = multivol_rng()
    N.check(N.lib().arcn_generate_grid_samples_multivol(
        density_grid.data_ptr(), int(density_grid_ema_step), int(n_elements), aabb_range.data_ptr(), int(n_cascade), int(n_grid),
        float(thresh), int(bool(inclusive)), rng.state, rng.inc, density_grid_positions_uniform.data_ptr(),
        density_grid_indices_uniform.data_ptr(), N.stream()), 'generate_grid_samples_multivol')
    rng.advance()


def update_bitfield_multivol(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid, n_cascade, inclusive):
    _chk(density_grid, density_grid_bitfield)
    _is(density_grid, torch.float32, 'density_grid')
    N.check(N.lib().arcn_update_bitfield_multivol(density_grid.data_ptr(), float(density_grid_mean), None,
                                                 density_grid_bitfield.data_ptr(), float(thres), int(n_grid), int(n_cascade),
                                                 int(bool(inclusive)), N.stream()), 'update_bitfield_multivol')
