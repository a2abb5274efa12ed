"""BitfieldBound (arcnerf/models/base_modules/obj_bound/bitfield_bound.py:15-189): the single-level instant-ngp bound — a
float density grid of n_grid^3 cells in Morton order, its packed bitfield, and the `_bitfield_func` kernels (K5-K10) that
sample through it and keep it current.

Differences from the reference, none of them observable: the grids are updated in place instead of being re-bound to the
op's return value, and the mean density stays on the device (the reference does `.item()` and rebuilds a 1-element tensor
that pybind converts back to a float: one host round trip per refresh)."""
import torch

from ....geometry.volume import Volume
from ....ops.bitfield_func import (CUDA_BACKEND_AVAILABLE, count_bitfield, ema_grid_samples_nerf, generate_grid_samples,
                                   sparse_volume_sampling_bit, splat_grid_samples, update_bitfield)
from ....utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


@BOUND_REGISTRY.register()
class BitfieldBound(BasicBound):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        assert valid_key_in_cfgs(cfgs, 'bitfield'), 'You must have bitfield in the cfgs'
        assert CUDA_BACKEND_AVAILABLE, 'bitfield requires libarcnerf_hip.so (there is no torch fallback)'
        bc = cfgs.bitfield
        if get_value_from_cfgs_field(bc, 'n_grid') is None:
            bc.n_grid = 128
        self.volume = Volume(**bc.__dict__)   # only its geometry (range, diagonal, ray test) is used
        self.n_grid = self.volume.get_n_grid()
        assert self.n_grid & (self.n_grid - 1) == 0 and 2 <= self.n_grid <= 1024, 'Morton layout: n_grid is a power of two <= 1024'
        self.n_elements = self.n_grid ** 3
        if self.get_optim_cfgs('epoch_optim') is not None:
            self.register_buffer('density_bitfield', torch.full((self.n_elements // 8,), 255, dtype=torch.uint8))
            self.register_buffer('density_grid', torch.zeros((self.n_elements,), dtype=torch.float32))
            self.register_buffer('density_grid_tmp', torch.zeros((self.n_elements,), dtype=torch.float32))
            self.ema_step = 0

    def get_obj_bound(self):
        return self.density_bitfield if self.get_optim_cfgs('epoch_optim') is not None else None

    def get_n_grid(self):
        return self.n_grid

    def read_optim_cfgs(self):
        p = super().read_optim_cfgs()
        p['near_distance'] = get_value_from_cfgs_field(self.cfgs, 'near_distance', 0.0)
        return p

    def get_near_far_from_rays(self, inputs, **kwargs):
        near, far, _, mask_rays = self.volume.ray_volume_intersection(inputs['rays_o'], inputs['rays_d'])
        return near, far, mask_rays[:, 0]

    def get_zvals_from_near_far(self, near, far, n_pts, inference_only=False, inverse_linear=False, perturb=False,
                                rays_o=None, rays_d=None, **kwargs):
        """zvals (B,n_pts), mask_pts (B,n_pts): marching through the Morton bitfield (K5) at a constant step"""
        dt = self.volume.get_diag_len() / n_pts
        return sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, dt, self.volume.get_range(), self.n_grid,
                                          self.density_bitfield, near_distance=self.get_optim_cfgs('near_distance'))

    def get_density_grid_mean(self):
        """(1,) device tensor: mean of the non-negative part of the grid"""
        return self.density_grid.clamp_min(0.0).mean().view(1)

    def get_bitfield_count(self, level=0):
        bitcount = count_bitfield(self.density_bitfield, self.n_grid)
        return bitcount, float(bitcount) / float(self.n_elements)

    @torch.no_grad()
    def optimize(self, cur_epoch=0, n_pts=128, get_est_opacity=None):
        """every `epoch_optim` steps: all cells during warm-up, then n/4 uniform + n/4 from occupied cells"""
        epoch_optim = self.get_optim_cfgs('epoch_optim')
        warmup = self.get_optim_cfgs('epoch_optim_warmup')
        if cur_epoch <= 0 or epoch_optim is None or cur_epoch % epoch_optim != 0:
            return
        if warmup is not None and cur_epoch < warmup:
            self._update_density_grid(self.n_elements, 0, get_est_opacity, n_pts)
        else:
            self._update_density_grid(self.n_elements // 4, self.n_elements // 4, get_est_opacity, n_pts)

    def _update_density_grid(self, n_uniform, n_nonuniform, get_est_opacity, n_pts):
        pos_u, idx_u = generate_grid_samples(self.density_grid, n_uniform, self.ema_step, self.n_grid, -0.01)
        pos_n, idx_n = generate_grid_samples(self.density_grid, n_nonuniform, self.ema_step, self.n_grid,
                                             self.get_optim_cfgs('opa_thres'))
        pos = torch.cat([pos_u, pos_n], dim=0)
        idx = torch.cat([idx_u, idx_n], dim=0)
        rng = self.volume.get_range()
        pos = pos * (rng[:, 1] - rng[:, 0])[None] + rng[:, 0][None]
        dt = self.volume.get_diag_len() / float(n_pts)
        opacity = get_est_opacity(dt, pos)
        self.density_grid_tmp.zero_()
        splat_grid_samples(opacity, idx, n_uniform + n_nonuniform, self.density_grid_tmp)
        ema_grid_samples_nerf(self.density_grid_tmp, self.density_grid, self.n_elements, self.get_optim_cfgs('ema_optim_decay'))
        update_bitfield(self.density_grid, self.get_density_grid_mean(), self.density_bitfield, self.get_optim_cfgs('opa_thres'),
                        self.n_grid)
        self.ema_step += 1
