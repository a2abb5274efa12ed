"""Morton-order density grid + packed bitfield shared by the single-level bound (BitfieldBound) and the cascade (MultiVol).

Both reference classes carry the same three buffers and the same refresh recipe (obj_bound/bitfield_bound.py:38-52,135-189 and
models/multivol_bkg_model.py:56-70,200-261): draw cells (K6 / K12), ask the model for their opacity, splat the per-cell maximum
(K7), decay-and-max into the running grid (K8), re-threshold into bits against min(opa_thres, mean) (K9).  Here that lives
once, as a mixin, so the buffers keep their reference names directly on the owning module (state_dict compatible).  The grids are
updated in place and the mean stays on the device."""
import torch

from ..ops.bitfield_func import ema_grid_samples_nerf, splat_grid_samples


class MortonDensityGrid:
    """mixin for an nn.Module; call `_alloc_density_grid` from __init__ after nn.Module.__init__"""

    def _alloc_density_grid(self, n_cells):
        assert n_cells % 8 == 0
        self.n_density_cells = int(n_cells)
        self.register_buffer('density_bitfield', torch.full((n_cells // 8,), 255, dtype=torch.uint8))  # everything occupied
        self.register_buffer('density_grid', torch.zeros((n_cells,), dtype=torch.float32))
        self.register_buffer('density_grid_tmp', torch.zeros((n_cells,), dtype=torch.float32))
        self.ema_step = 0

    def get_density_grid_mean(self):
        """(1,) device tensor: mean of the non-negative part of the grid (all levels together)"""
        return self.density_grid.clamp_min(0.0).mean().view(1)

    @staticmethod
    def refresh_plan(cur_epoch, every, warmup, n_cells):
        """None when `cur_epoch` is not a refresh step, else (n_uniform, n_from_occupied): every cell during warm-up, a quarter
        of the cells uniformly plus a quarter drawn among the occupied ones afterwards"""
        if cur_epoch <= 0 or every is None or cur_epoch % every != 0:
            return None
        if warmup is not None and cur_epoch < warmup:
            return n_cells, 0
        return n_cells // 4, n_cells // 4

    def _refresh_density_grid(self, plan, draw, opacity_of, decay, opa_thres, repack):
        """draw(n, thresh) -> world positions (n,3), cell indices (n,) int32; opacity_of(positions) -> (n,);
        repack(grid, mean, bits) thresholds the grid into the bitfield"""
        n_uniform, n_occupied = plan
        pos_u, idx_u = draw(n_uniform, -0.01)          # any cell: every density passes -0.01
        pos_o, idx_o = draw(n_occupied, opa_thres)     # biased towards cells above the threshold
        opacity = opacity_of(torch.cat([pos_u, pos_o], dim=0))
        self.density_grid_tmp.zero_()
        splat_grid_samples(opacity, torch.cat([idx_u, idx_o], dim=0), n_uniform + n_occupied, self.density_grid_tmp)
        ema_grid_samples_nerf(self.density_grid_tmp, self.density_grid, self.n_density_cells, decay)
        repack(self.density_grid, self.get_density_grid_mean(), self.density_bitfield)
        self.ema_step += 1
