"""MultiVol (arcnerf/models/multivol_bkg_model.py:18-261): instant-ngp's cascade of nested volumes, as a background model
(inner volume excluded) or as a whole-scene model (`inclusive`).  Volume m is the basic volume scaled 2^m about its origin; each
level is an n_grid^3 Morton density grid + bitfield, stored level after level.  Sampling = the cone-stepping cascade marcher
(K11); pruning = the shared Morton-grid refresh with the cascade's cell draw (K12) and bit packing.  SURVEY.md section 8f, rank 2."""
import torch

from ..geometry.density_grid import MortonDensityGrid
from ..geometry.ray import aabb_ray_intersection
from ..geometry.volume import Volume
from ..ops import functional as Fn
from ..ops.autograd import PackedCompositeFn
from ..ops.multivol_func import (CUDA_BACKEND_AVAILABLE, generate_grid_samples_multivol, multivol_rng,
                                 sparse_sampling_in_multivol_bitfield, update_bitfield_multivol)
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.registry import MODEL_REGISTRY
from ..utils.torch_utils import chunk_processing
from .base_modules import build_geo_model, build_radiance_model
from .bkg_model import BkgModel
from .masked_samples import hold_rays, nets_on_valid_samples


@MODEL_REGISTRY.register()
class MultiVol(BkgModel, MortonDensityGrid):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        assert CUDA_BACKEND_AVAILABLE, 'multivol requires libarcnerf_hip.so (there is no torch fallback)'
        self.geo_net = build_geo_model(self.cfgs.model.geometry)
        self.radiance_net = build_radiance_model(self.cfgs.model.radiance)
        vc = self.cfgs.model.basic_volume
        self.n_grid = get_value_from_cfgs_field(vc, 'n_grid', 128)
        self.n_cascade = int(vc.n_cascade)
        assert self.n_cascade > 1, 'You should have at least 2 cascades...'
        self.inclusive = bool(get_value_from_cfgs_field(vc, 'inclusive', False))
        box = {k: v for k, v in vc.__dict__.items() if k in ('origin', 'side', 'xyz_len')}
        self.basic_volume = Volume(n_grid=self.n_grid, **box)
        grow = 2 ** (self.n_cascade - 1)
        self.max_volume = Volume(origin=self.basic_volume.get_origin(), xyz_len=[s * grow for s in self.basic_volume.get_len()])
        # step length: from one coarse-sample interval of the inner volume up to one cell of the outermost one
        self.cone_angle = get_value_from_cfgs_field(self.cfgs.model.rays, 'cone_angle', 0.0)
        self.min_step = self.basic_volume.get_diag_len() / self.get_ray_cfgs('n_sample')
        self.max_step = self.max_volume.get_diag_len() / self.n_grid
        self.n_elements = self.n_grid ** 3
        self.n_levels = self.n_cascade if self.inclusive else self.n_cascade - 1
        self.total_n_elements = self.n_elements * self.n_levels
        self._alloc_density_grid(self.total_n_elements)
        self.use_packed_path = True   # set False to force the dense reference-shaped path (always taken when progress is requested)

    def get_near_far_from_rays(self, rays_o, rays_d):
        """near, far (B,1) against the outermost volume, torch semantics: the cameras sit inside it"""
        box = self.max_volume.get_range()[None].to(rays_o.device)
        near, far, _, _ = aabb_ray_intersection(rays_o, rays_d, box, force_torch=True, want_pts=False)
        return near, far

    def get_zvals_from_near_far(self, near, far, n_pts, rays_o, rays_d, **kwargs):
        """zvals (B, n_pts), mask_pts (B, n_pts) from the cascade marcher"""
        return sparse_sampling_in_multivol_bitfield(rays_o, rays_d, near, far, n_pts, self.cone_angle, self.min_step, self.max_step,
                                                    self.basic_volume.get_range(), self.max_volume.get_range(), self.n_grid,
                                                    self.n_cascade, self.density_bitfield,
                                                    near_distance=self.get_optim_cfgs('near_distance'), inclusive=self.inclusive)

    def get_sigma_radiance_by_mask_pts(self, geo_net, radiance_net, rays_o, rays_d, zvals, mask_pts):
        return nets_on_valid_samples(self._forward_pts_dir, self.chunk_pts, geo_net, radiance_net, rays_o, rays_d, zvals, mask_pts)

    @torch.no_grad()
    def _sample_begin(self, rays_o, rays_d):
        """bounds + cascade marcher + scan queued, the sample total on its way to the host (no wait here)"""
        n_pts = self.get_ray_cfgs('n_sample')
        near, far = self.get_near_far_from_rays(rays_o, rays_d)
        rng = multivol_rng()
        zvals, _, counts = Fn.sparse_sampling_in_multivol_bitfield(
            rays_o, rays_d, near, far, n_pts, self.cone_angle, self.min_step, self.max_step,
            self.basic_volume.get_range23(), self.max_volume.get_range23(),
            self.n_grid, self.n_cascade, self.density_bitfield, self.get_optim_cfgs('near_distance'), self.inclusive, rng.state,
            rng.inc, want_counts=True, dense=False)
        rng.advance()
        h = Fn.pack_dense_samples_begin(zvals, counts)
        hold_rays(h, rays_o, rays_d)
        return h

    def presample(self, inputs):
        """FullModel calls this BEFORE the foreground model runs (same rays): the background's sampler is queued and its sample count
        travels to the host while the foreground marches and waits for ITS count - one wait for the two host reads of a step instead
        of two.  The sampler has its own generator (ops.multivol_func.multivol_rng), so the order of the two samplers does not change
        either stream.  Used by the next forward() on the same ray tensors, dropped otherwise."""
        rays_o, rays_d = inputs['rays_o'], inputs['rays_d']
        if not (self.use_packed_path and rays_o.is_cuda and rays_o.is_contiguous() and rays_d.is_contiguous()
                and rays_o.dtype == torch.float32 and rays_d.dtype == torch.float32):
            return
        self._presampled = ((rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], rays_o._version, rays_d._version),
                            self._sample_begin(rays_o, rays_d))

    def _forward_packed(self, rays_o, rays_d, inference_only):
        """Same result as the dense path below when only rgb / depth / mask are wanted, without the padded (rays, slots) tensors:
        the sampler's per-ray counts are scanned into offsets, the valid samples compacted (kernels, one host read for the total),
        the nets see the packed points and the packed compositor does the rest - no boolean-mask gathers or scatters."""
        n_rays = rays_o.shape[0]
        pre, self._presampled = getattr(self, '_presampled', None), None
        if pre is None or pre[0] != (rays_o.data_ptr(), rays_d.data_ptr(), n_rays, rays_o._version, rays_d._version):
            pre = (None, self._sample_begin(rays_o, rays_d))
        else:
            # (possibly queued on the sampling stream a step ago - FullModel.prefetch_samples: order this stream behind it and keep its
            # tensors from being handed out again while this stream still reads them)
            cur = torch.cuda.current_stream()
            cur.wait_event(pre[1]['event'])
            for t_ in pre[1].values():
                if isinstance(t_, torch.Tensor) and t_.is_cuda:
                    t_.record_stream(cur)
        with torch.no_grad():
            t, ray_id, offsets, p_dense, total = Fn.pack_dense_samples_end(pre[1])
            if total > 0:
                xyz, dirs = Fn.packed_points(rays_o, rays_d, t, ray_id)
        if total == 0:   # nothing sampled anywhere: empty rays composite to 0 (+ white background)
            zero = rays_o.new_zeros((n_rays,))
            rgb = rays_o.new_ones((n_rays, 3)) if self.get_ray_cfgs('white_bkg') else rays_o.new_zeros((n_rays, 3))
            return {'rgb': rgb, 'depth': zero, 'mask': zero.clone()}
        sigma, radiance = chunk_processing(self._forward_pts_dir, self.chunk_pts, False, self.geo_net, self.radiance_net, xyz, dirs)
        noise_std = float(self.get_ray_cfgs('noise_std') or 0.0) if not inference_only else 0.0
        noise = torch.randn_like(sigma) * noise_std if noise_std > 0 else None
        rgb, depth, mask = PackedCompositeFn.apply(sigma.contiguous(), radiance.contiguous(), t, offsets, p_dense, bool(self.add_inf_z),
                                                   bool(self.get_ray_cfgs('white_bkg')), noise)
        return {'rgb': rgb, 'depth': depth, 'mask': mask}

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d = inputs['rays_o'], inputs['rays_d']
        if self.use_packed_path and not get_progress and rays_o.is_cuda:
            return self._forward_packed(rays_o.contiguous().float(), rays_d.contiguous().float(), inference_only)
        with torch.no_grad():
            near, far = self.get_near_far_from_rays(rays_o, rays_d)
            zvals, mask_pts = self.get_zvals_from_near_far(near, far, self.get_ray_cfgs('n_sample'), rays_o, rays_d)
            width = max(1, int(mask_pts.sum(dim=1).max()))   # the longest ray decides the padded width
            zvals, mask_pts = zvals[:, :width].contiguous(), mask_pts[:, :width].contiguous()
        sigma, radiance = self.get_sigma_radiance_by_mask_pts(self.geo_net, self.radiance_net, rays_o, rays_d, zvals, mask_pts)
        return self.output_get_progress(self.ray_marching(sigma, radiance, zvals, inference_only=inference_only), get_progress)

    @torch.no_grad()
    def optimize(self, cur_epoch=0):
        plan = self.refresh_plan(cur_epoch, self.get_optim_cfgs('epoch_optim'), self.get_optim_cfgs('epoch_optim_warmup'),
                                 self.total_n_elements)
        if plan is not None:
            self._update_density_grid(*plan, self.get_ray_cfgs('n_sample'))

    def _update_density_grid(self, n_uniform, n_nonuniform, n_pts):
        inner = self.basic_volume.get_range()
        dt = self.basic_volume.get_diag_len() / float(n_pts)   # opacity over one inner-volume step, whatever the level

        def draw(n, thresh):
            return generate_grid_samples_multivol(self.density_grid, n, inner, self.ema_step, self.n_cascade, self.n_grid, thresh,
                                                  self.inclusive)

        def repack(grid, mean, bits):
            update_bitfield_multivol(grid, mean, bits, self.get_optim_cfgs('opa_thres'), self.n_grid, self.n_cascade, self.inclusive)

        self._refresh_density_grid((n_uniform, n_nonuniform), draw, lambda pos: self.get_est_opacity(dt, pos),
                                   self.get_optim_cfgs('ema_optim_decay'), self.get_optim_cfgs('opa_thres'), repack)
