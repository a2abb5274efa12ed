"""BitfieldBound (arcnerf/models/base_modules/obj_bound/bitfield_bound.py:15-189): the single-level instant-ngp bound — a
float density grid of n_grid^3 cells in Morton order, its packed bitfield, and the `_bitfield_func` kernels (K5-K10) that
sample through it and keep it current.  The grid state and its refresh recipe are the shared MortonDensityGrid mixin."""
import torch

from ....geometry.density_grid import MortonDensityGrid
from ....geometry.volume import Volume
from ....ops.bitfield_func import CUDA_BACKEND_AVAILABLE, count_bitfield, generate_grid_samples, sparse_volume_sampling_bit, update_bitfield
from ....utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


@BOUND_REGISTRY.register()
class BitfieldBound(BasicBound, MortonDensityGrid):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        assert valid_key_in_cfgs(cfgs, 'bitfield'), 'You must have bitfield in the cfgs'
        assert CUDA_BACKEND_AVAILABLE, 'bitfield requires libarcnerf_hip.so (there is no torch fallback)'
        box = dict(cfgs.bitfield.__dict__)
        box['n_grid'] = get_value_from_cfgs_field(cfgs.bitfield, 'n_grid', 128)
        self.volume = Volume(**box)   # geometry only: range, diagonal, ray / box test
        self.n_grid = self.volume.get_n_grid()
        assert self.n_grid & (self.n_grid - 1) == 0 and 2 <= self.n_grid <= 1024, 'Morton layout: n_grid is a power of two <= 1024'
        self.n_elements = self.n_grid ** 3
        self.pruning = self.get_optim_cfgs('epoch_optim') is not None
        if self.pruning:
            self._alloc_density_grid(self.n_elements)

    def read_optim_cfgs(self):
        p = super().read_optim_cfgs()
        p['near_distance'] = get_value_from_cfgs_field(self.cfgs, 'near_distance', 0.0)
        return p

    def get_obj_bound(self):
        return self.density_bitfield if self.pruning else None

    def get_n_grid(self):
        return self.n_grid

    def get_bitfield_count(self, level=0):
        """(count, count / n_cells) with the reference's counting rule (ops/bitfield_func.py:196-206)"""
        n = count_bitfield(self.density_bitfield, self.n_grid)
        return n, float(n) / float(self.n_elements)

    def get_near_far_from_rays(self, inputs, **kwargs):
        near, far, _, hit = self.volume.ray_volume_intersection(inputs['rays_o'], inputs['rays_d'])
        return near, far, hit[:, 0]

    def get_zvals_from_near_far(self, near, far, n_pts, inference_only=False, inverse_linear=False, perturb=False,
                                rays_o=None, rays_d=None, **kwargs):
        """zvals (B,n_pts), mask_pts (B,n_pts): constant-step marching through the Morton bitfield (K5)"""
        step = self.volume.get_diag_len() / n_pts
        return sparse_volume_sampling_bit(rays_o, rays_d, near, far, n_pts, step, self.volume.get_range(), self.n_grid,
                                          self.density_bitfield, near_distance=self.get_optim_cfgs('near_distance'))

    @torch.no_grad()
    def optimize(self, cur_epoch=0, n_pts=128, get_est_opacity=None):
        plan = self.refresh_plan(cur_epoch, self.get_optim_cfgs('epoch_optim'), self.get_optim_cfgs('epoch_optim_warmup'),
                                 self.n_elements)
        if plan is not None:
            self._update_density_grid(*plan, get_est_opacity, n_pts)

    def _update_density_grid(self, n_uniform, n_nonuniform, get_est_opacity, n_pts):
        box = self.volume.get_range()
        lo, extent = box[:, 0][None], (box[:, 1] - box[:, 0])[None]
        dt = self.volume.get_diag_len() / float(n_pts)

        def draw(n, thresh):   # K6 draws in the unit cube; move the points into the volume
            unit, cell = generate_grid_samples(self.density_grid, n, self.ema_step, self.n_grid, thresh)
            return unit * extent + lo, cell

        def repack(grid, mean, bits):
            update_bitfield(grid, mean, bits, self.get_optim_cfgs('opa_thres'), self.n_grid)

        self._refresh_density_grid((n_uniform, n_nonuniform), draw, lambda pos: get_est_opacity(dt, pos),
                                   self.get_optim_cfgs('ema_optim_decay'), self.get_optim_cfgs('opa_thres'), repack)
