"""VolumeBound (arcnerf/models/base_modules/obj_bound/volume_bound.py:15-212): an occupancy-pruned volume around the
object.  near/far = ray/AABB kernel (K2), zvals = occupancy marching kernel (K3), optimize = periodic occupancy refresh.

optimize(): the reference does unique(voxel_idx) + segmented max (K4) + indexed EMA update; here one scatter-max kernel
and one grid pass give the same opacity field without the sort (tests pin the equivalence against the reference's
golden update).
"""
import torch

import numpy as np

from ....geometry.volume import Volume, refresh_tape, select_refresh_cells
from ....ops import volume_func as _vf
from ....ops.volume_func import sparse_volume_sampling
from ....render.ray_helper import get_zvals_from_near_far_fix_step, handle_valid_mask_zvals
from ....utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


@BOUND_REGISTRY.register()
class VolumeBound(BasicBound):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        assert valid_key_in_cfgs(cfgs, 'volume'), 'You must have volume in the cfgs'
        vc = cfgs.volume
        if get_value_from_cfgs_field(vc, 'n_grid') is None:
            vc.n_grid = 128
        self.volume = Volume(**vc.__dict__)
        if self.get_optim_cfgs('epoch_optim') is not None:
            self.volume.set_up_voxel_bitfield(init_occ=True)
            self.volume.set_up_voxel_opafield()

    def get_obj_bound(self):
        return self.volume

    def read_optim_cfgs(self):
        p = super().read_optim_cfgs()
        p['ray_sample_acc'] = get_value_from_cfgs_field(self.cfgs, 'ray_sample_acc', False)
        p['ray_sample_fix_step'] = get_value_from_cfgs_field(self.cfgs, 'ray_sample_fix_step', False)
        p['near_distance'] = get_value_from_cfgs_field(self.cfgs, 'near_distance', 0.0)
        return p

    def uses_sparse_sampling(self):
        return self.get_optim_cfgs('epoch_optim') is not None and bool(self.get_optim_cfgs('ray_sample_acc'))

    def get_near_far_from_rays(self, inputs, **kwargs):
        near, far, _, mask_rays = self.volume.ray_volume_intersection(inputs['rays_o'], inputs['rays_d'])
        return near, far, mask_rays[:, 0]

    def get_zvals_from_near_far(self, near, far, n_pts, inference_only=False, inverse_linear=False, perturb=False,
                                rays_o=None, rays_d=None, **kwargs):
        if self.uses_sparse_sampling():
            return self.get_zvals_from_sparse_volume(rays_o, rays_d, near, far, n_pts, inference_only, inverse_linear, perturb)
        return super().get_zvals_from_near_far(near, far, n_pts, inference_only, inverse_linear, perturb)

    @torch.no_grad()
    def get_zvals_from_sparse_volume(self, rays_o, rays_d, near, far, n_pts, inference_only, inverse_linear, perturb):
        """zvals (B,n_pts), mask_pts (B,n_pts) [T..T F..F], tails repeat the last valid z (volume_bound.py:95-143).

        The occupancy marcher (K3) whenever the native module is there - which on this platform is always, the product path has no
        CPU branch.  `ops.volume_func.CUDA_BACKEND_AVAILABLE = False` selects what the reference does WITHOUT its extension
        (volume_bound.py:126-141, SURVEY a4'): fixed-step (or uniform) zvals, the occupancy test of every sample (`check_pts_in_occ_voxel`
        - here still the HIP kernel, on GPU tensors), stable compaction of the survivors.  A different algorithm from K3 (no voxel
        skipping, no start jitter; it cannot even run at n_grid 128 on the reference's CPU path: 293 GB), kept so that a caller who
        compares against an extension-less reference has the same semantics."""
        if _vf.CUDA_BACKEND_AVAILABLE:
            dt = self.volume.get_diag_len() / n_pts
            return sparse_volume_sampling(rays_o, rays_d, near, far, n_pts, dt, self.volume.get_range(), self.volume.get_n_grid(),
                                          self.volume.get_voxel_bitfield(), near_distance=self.get_optim_cfgs('near_distance'))
        if self.get_optim_cfgs('ray_sample_fix_step'):
            zvals, mask_pts = self.get_zvals_from_near_far_fix_step(near, far, n_pts, inference_only, perturb)
            pts = rays_o[:, None, :] + zvals[..., None] * rays_d[:, None, :]
            valid = self.volume.check_pts_in_occ_voxel(pts[mask_pts].view(-1, 3))
            mask_pts = mask_pts.clone()
            mask_pts[mask_pts.clone()] = valid
        else:
            zvals, _ = super().get_zvals_from_near_far(near, far, n_pts, inference_only, inverse_linear, perturb)
            pts = rays_o[:, None, :] + zvals[..., None] * rays_d[:, None, :]
            mask_pts = self.volume.check_pts_in_occ_voxel(pts.view(-1, 3)).view(-1, n_pts)
        return handle_valid_mask_zvals(zvals, mask_pts)

    def get_zvals_from_near_far_fix_step(self, near, far, n_pts, inference_only=False, perturb=False, **kwargs):
        """fixed step = diagonal / n_pts (volume_bound.py:145-158)"""
        fix_t = self.volume.get_diag_len() / n_pts
        return get_zvals_from_near_far_fix_step(near, far, fix_t, n_pts, perturb=perturb if not inference_only else False)

    @torch.no_grad()
    def optimize(self, cur_epoch=0, n_pts=128, get_est_opacity=None):
        epoch_optim = self.get_optim_cfgs('epoch_optim')
        warmup = self.get_optim_cfgs('epoch_optim_warmup')
        if cur_epoch <= 0 or epoch_optim is None or cur_epoch % epoch_optim != 0:
            return
        vol = self.volume
        n = vol.get_n_grid()
        dev = vol.get_device()
        n_dev = None
        pts = None
        tape = refresh_tape()
        perm = uni = None
        if tape is not None:    # the draws of a recorded run instead of the seeded generators (geometry/volume.py:set_refresh_tape)
            perm, uni = tape.draws(cur_epoch, vol.get_n_voxel(), dev)
        if warmup is not None and cur_epoch < warmup:
            cell = torch.arange(vol.get_n_voxel(), device=dev)
        else:
            if not hasattr(self, '_refresh_cache'):
                self._refresh_cache, self._refresh_rng = {}, np.random.default_rng(12345)
            bits = vol.get_voxel_bitfield(flatten=True)
            if tape is not None:
                cell, n_dev = select_refresh_cells(bits, vol.get_n_voxel(), self._refresh_cache, self._refresh_rng, perm=perm)
            elif bits.is_cuda and bits.is_contiguous() and bits.data_ptr() % 8 == 0 and n >= 16 and n & (n - 1) == 0:
                # cells (n / 4 uniform along the Z-curve + the first n / 4 occupied) and their jittered points in four small launches
                # (arcn_refresh_cells_points; the torch formulation below is ~45)
                from ....geometry.volume import mix_constants
                from ....ops import functional as Fn
                nc, rng, ch = vol.get_n_voxel(), self._refresh_rng, self._refresh_cache
                if 'native' not in ch:
                    ch['native'] = {'cells': torch.zeros(2 * (nc // 4), dtype=torch.int64, device=dev),
                                    'pts': torch.zeros((2 * (nc // 4), 3), dtype=torch.float32, device=dev),
                                    'n_valid': torch.zeros(1, dtype=torch.int32, device=dev),
                                    'ws': torch.empty(nc + 8 * (nc // 4096 + 2), dtype=torch.uint8, device=dev)}
                rb = ch['native']
                # host copies of the voxel size and the lower corner: read once (one synchronisation) unless the volume is learnable
                fixed = not (vol.origin.requires_grad or vol.xyz_len.requires_grad)
                if not fixed or ch.get('geom_n') != n:
                    ch['geom'] = (vol.get_voxel_size(), vol.get_range()[:, 0].tolist())
                    ch['geom_n'] = n
                vsz, mn = ch['geom']
                Fn.refresh_cells_points(bits, n, mix_constants(nc, rng), vsz, mn,
                                        int(rng.integers(0, 1 << 62)), int(rng.integers(0, 1 << 62)) * 2 + 1, rb['cells'], rb['pts'], rb['n_valid'],
                                        rb['ws'])
                cell, pts, n_dev = rb['cells'], rb['pts'], rb['n_valid']
            else:
                # entries past n_dev are stale cells from an earlier refresh: evaluated by the net (harmless) and skipped by the
                # scatter, which honours the device-side count
                cell, n_dev = select_refresh_cells(bits, vol.get_n_voxel(), self._refresh_cache, self._refresh_rng)
        if pts is None:
            pts = vol.get_voxel_pts_by_voxel_idx(vol.convert_flatten_index_to_xyz_index(cell, n).float())
            noise = (torch.rand_like(pts) if uni is None else uni[:pts.shape[0]]) - 0.5
            pts = pts + noise * vol.get_voxel_size(to_list=False)[None, :]
        dt = vol.get_diag_len() / float(n_pts)
        opacity = get_est_opacity(dt, pts.contiguous())
        vol.update_opafield_by_flat_idx(cell, opacity, ema=self.get_optim_cfgs('ema_optim_decay'), n_dev=n_dev)
        vol.update_bitfield_by_opafield(threshold=self.get_optim_cfgs('opa_thres'), ops='overwrite')
