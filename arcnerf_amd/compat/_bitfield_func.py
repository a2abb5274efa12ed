"""`_bitfield_func` with the pybind signatures of arcnerf/ops/src/bitfield_func/bitfield_func.cpp:279-286: outputs are
caller-allocated tensors written in place, nothing is returned.  RuntimeError on bad inputs like CHECK_INPUT."""
import torch

from arcnerf_amd import _native as N
from arcnerf_amd.ops.bitfield_func import bitfield_rng


def _chk(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('input must be a CUDA tensor')
        if not t.is_contiguous():
            raise RuntimeError('input must be contiguous')


def _is(t, dtype, what):
    if t.dtype != dtype:
        raise RuntimeError('{} must be {}'.format(what, dtype))


def sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, aabb_range, n_grid, bitfield, near_distance, zvals, mask):
    _chk(rays_o, rays_d, near, far, aabb_range, bitfield, zvals, mask)
    for t, w in ((rays_o, 'rays_o'), (rays_d, 'rays_d'), (near, 'near'), (far, 'far'), (aabb_range, 'aabb_range'), (zvals, 'zvals')):
        _is(t, torch.float32, w)
    _is(bitfield, torch.uint8, 'bitfield')
    _is(mask, torch.bool, 'mask')
    if rays_o.shape[1] != 3 or rays_d.shape[1] != 3:
        raise RuntimeError('Input rays tensor must be (B, 3).')
    if near.shape[1] != 1 or far.shape[1] != 1:
        raise RuntimeError('Input near/far tensor must be (B, 1).')
    if tuple(aabb_range.shape) != (2, 3):
        raise RuntimeError('xyz range should be in (2, 3).')
    if bitfield.shape[0] != n_grid * n_grid * n_grid // 8:
        raise RuntimeError('bitfield should be in (n_grid**3/8,).')
    if tuple(zvals.shape) != (rays_o.shape[0], n_pts) or tuple(mask.shape) != (rays_o.shape[0], n_pts):
        raise RuntimeError('zval / mask should be in (n_rays, n_pts).')
    rng = bitfield_rng()
    N.check(N.lib().arcn_sparse_volume_sampling_bit(rays_o.data_ptr(), rays_d.data_ptr(), near.data_ptr(), far.data_ptr(),
                                                   int(n_pts), float(dt), aabb_range.data_ptr(), int(n_grid),
                                                   bitfield.data_ptr(), float(near_distance), rng.state, rng.inc,
                                                   zvals.data_ptr(), mask.data_ptr(), None, rays_o.shape[0], N.stream()),
            'sparse_volume_sampling_bit')
    rng.advance()


def generate_grid_samples(density_grid, density_grid_ema_step, n_elements, n_grid, thresh, density_grid_positions_uniform,
                          density_grid_indices_uniform):
    _chk(density_grid, density_grid_positions_uniform, density_grid_indices_uniform)
    _is(density_grid, torch.float32, 'density_grid')
    _is(density_grid_positions_uniform, torch.float32, 'positions')
    _is(density_grid_indices_uniform, torch.int32, 'indices')
    rng = bitfield_rng()
    N.check(N.lib().arcn_generate_grid_samples(density_grid.data_ptr(), int(density_grid_ema_step), int(n_elements), int(n_grid),
                                              float(thresh), rng.state, rng.inc, density_grid_positions_uniform.data_ptr(),
                                              density_grid_indices_uniform.data_ptr(), N.stream()), 'generate_grid_samples')
    rng.advance()


def splat_grid_samples(density, density_grid_indices, n_density_grid_samples, density_grid_tmp):
    _chk(density, density_grid_indices, density_grid_tmp)
    _is(density, torch.float32, 'density')
    _is(density_grid_indices, torch.int32, 'density_grid_indices')
    _is(density_grid_tmp, torch.float32, 'density_grid_tmp')
    N.check(N.lib().arcn_splat_grid_samples(density.data_ptr(), density_grid_indices.data_ptr(), int(n_density_grid_samples),
                                           density_grid_tmp.data_ptr(), N.stream()), 'splat_grid_samples')


def ema_grid_samples_nerf(density_grid_tmp, n_elements, decay, density_grid):
    _chk(density_grid_tmp, density_grid)
    _is(density_grid_tmp, torch.float32, 'density_grid_tmp')
    _is(density_grid, torch.float32, 'density_grid')
    N.check(N.lib().arcn_ema_grid_samples_nerf(density_grid_tmp.data_ptr(), int(n_elements), float(decay),
                                              density_grid.data_ptr(), N.stream()), 'ema_grid_samples_nerf')


def update_bitfield(density_grid, density_grid_mean, density_grid_bitfield, thres, n_grid):
    _chk(density_grid, density_grid_bitfield)
    _is(density_grid, torch.float32, 'density_grid')
    N.check(N.lib().arcn_update_bitfield(density_grid.data_ptr(), float(density_grid_mean), None, density_grid_bitfield.data_ptr(),
                                        float(thres), int(n_grid), N.stream()), 'update_bitfield')


def count_bitfield(density_grid_bitfield, counter, n_grid):
    N.check(N.lib().arcn_count_bitfield(density_grid_bitfield.data_ptr(), counter.data_ptr(), int(n_grid), N.stream()),
            'count_bitfield')
