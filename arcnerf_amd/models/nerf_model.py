"""NeRF model (arcnerf/models/nerf_model.py:13-117): coarse (+ optional fine) geometry/radiance nets over the samples
of FgModel, hierarchical up-sampling by inverse-CDF.

MI355X fast path: when the configuration is the instant-ngp one (volume bound with occupancy marching, hash-grid encoder,
fused geo/radiance MLPs, SH view encoding, no hierarchical stage) `forward` bypasses the dense (R, 1024) tensors and the
host decisions of FgModel.forward and runs the packed, device-count-driven kernel sequence of arcnerf_amd.pipeline on the
module's own parameters.  Outputs (keys, shapes, values) are the same; tests compare both paths.
"""

import torch
from torch.autograd.function import once_differentiable

from ..ops.autograd import direct_grad
from ..pipeline import NgpConfig, NgpField, NgpPipeline
from ..render.ray_helper import sample_pdf
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.registry import MODEL_REGISTRY
from .base_modules import build_geo_model, build_radiance_model
from .base_modules.encoding import HashGridEmbedder, SHEmbedder
from .base_modules.geo_rad_model import FusedMLPGeoNet, FusedMLPRadianceNet
from .base_modules.obj_bound import BitfieldBound, VolumeBound
from .fg_model import FgModel


class _PackedRenderFn(torch.autograd.Function):
    """rgb, depth, mask = packed NGP render(rays; table, geo weights, radiance weights) with hand-written backward."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, bkg, table, geo_w, rad_w, pipe, train, noise_std):
        pipe.bind_params({'table': table.detach().view(-1), 'geo_w': geo_w.detach(), 'rad_w': rad_w.detach()})
        noise = pipe.buf['noise'].normal_(0.0, noise_std) if (train and noise_std > 0) else None
        # the compositor writes the per-ray outputs and nothing reads them back: hand it a fresh allocation (no launch) for this call
        # instead of cloning the pipeline's own buffers afterwards (three copy kernels per call)
        R, b = rays_o.shape[0], pipe.buf
        o = torch.empty(5 * R, dtype=torch.float32, device=rays_o.device)
        own = (b['rgb'], b['depth'], b['mask'])
        b['rgb'], b['depth'], b['mask'] = o[:3 * R].view(R, 3), o[3 * R:4 * R], o[4 * R:]
        try:
            rgb, depth, mask = pipe.forward(rays_o, rays_d, bkg, train=train, noise=noise, presampled=True)
        finally:
            b['rgb'], b['depth'], b['mask'] = own
        ctx.pipe, ctx.gen = pipe, pipe.generation
        ctx.save_for_backward(rays_o, rays_d, table, geo_w, rad_w)
        ctx.set_materialize_grads(False)   # an output the loss does not read arrives as None, not as a zero-filled tensor
        counts = b['counts'][:R].clone()
        ctx.mark_non_differentiable(counts)
        return rgb, depth, mask, counts

    @staticmethod
    @once_differentiable
    def backward(ctx, d_rgb, d_depth, d_mask, _):
        pipe = ctx.pipe
        if pipe.generation != ctx.gen:
            raise RuntimeError('the packed NGP path keeps ONE forward per backward: raise model.chunk_rays above the number '
                               'of rays of a training step (the reference default 32768 does) or call backward per chunk')
        rays_o, rays_d, table, geo_w, rad_w = ctx.saved_tensors
        # The kernels ACCUMULATE into the gradient buffers they are given.  A parameter whose .grad is a view of FusedAdam's flat gradient
        # buffer (optim.FusedAdam.flatten marks it) gets its gradient added there directly: no zero-filled temporary, no AccumulateGrad
        # pass (48.8 MB written, read and added again for the table) - the node then returns None for that input.
        # (direct_grad asks the engine whether THIS run accumulates into .grad: torch.autograd.grad(...) must not touch it)
        direct = {k: direct_grad(t) is not None for k, t in (('table', table), ('geo_w', geo_w), ('rad_w', rad_w))}
        g = {'table': (table.grad if direct['table'] else torch.zeros_like(table)).view(-1),
             'geo_w': geo_w.grad if direct['geo_w'] else torch.zeros_like(geo_w),
             'rad_w': rad_w.grad if direct['rad_w'] else torch.zeros_like(rad_w)}
        pipe.bind_grads(g)
        if d_rgb is None:
            d_rgb = torch.zeros((rays_o.shape[0], 3), dtype=torch.float32, device=rays_o.device)
        pipe.backward(rays_o, rays_d, d_rgb.contiguous(), None if d_depth is None else d_depth.contiguous(),
                      None if d_mask is None else d_mask.contiguous())
        pipe.bind_grads(None)
        return (None, None, None, None if direct['table'] else g['table'].view_as(table), None if direct['geo_w'] else g['geo_w'],
                None if direct['rad_w'] else g['rad_w'], None, None, None)


@MODEL_REGISTRY.register()
class NeRF(FgModel):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        self.coarse_geo_net = build_geo_model(self.cfgs.model.geometry)
        self.coarse_radiance_net = build_radiance_model(self.cfgs.model.radiance)
        self.ray_cfgs['n_importance'] = get_value_from_cfgs_field(self.cfgs.model.rays, 'n_importance', 0)
        self.ray_cfgs['shared_network'] = get_value_from_cfgs_field(self.cfgs.model.rays, 'shared_network', False)
        if self.get_ray_cfgs('n_importance') > 0:
            if self.get_ray_cfgs('shared_network'):
                self.fine_geo_net, self.fine_radiance_net = self.coarse_geo_net, self.coarse_radiance_net
            else:
                self.fine_geo_net = build_geo_model(self.cfgs.model.geometry)
                self.fine_radiance_net = build_radiance_model(self.cfgs.model.radiance)
        self.use_packed_path = True  # set False to force the dense reference-shaped path
        self._pipe = None

    def get_net(self):
        if self.get_ray_cfgs('n_importance') > 0:
            return self.fine_geo_net, self.fine_radiance_net
        return self.coarse_geo_net, self.coarse_radiance_net

    def init_setting(self):
        self.coarse_geo_net.pretrain_siren()
        if self.get_ray_cfgs('n_importance') > 0:
            self.fine_geo_net.pretrain_siren()

    # ---- packed instant-ngp path -------------------------------------------------------------------------
    def packed_path_eligible(self):
        g, r, b = self.coarse_geo_net, self.coarse_radiance_net, self.obj_bound
        pruned = (isinstance(b, VolumeBound) and b.uses_sparse_sampling()) or \
                 (isinstance(b, BitfieldBound) and b.get_obj_bound() is not None)
        return (self.use_packed_path and self.get_ray_cfgs('n_importance') == 0 and pruned and isinstance(g, FusedMLPGeoNet) and isinstance(g.embed_fn, HashGridEmbedder)
                and not g.embed_fn.include_input and g.W_feat > 0 and g.out_act is not None
                and type(g.out_act).__name__ == 'TruncExp' and isinstance(r, FusedMLPRadianceNet) and r.mode in ('fv', 'vf')
                and isinstance(r.embed_fn_view, SHEmbedder) and not r.embed_fn_view.include_input
                and self.get_ray_cfgs('near') is None and self.get_ray_cfgs('far') is None)

    def _packed_pipeline(self, device, min_samples=0):
        if self._pipe is not None and self._pipe.cap < min_samples:
            self._pipe = None   # grow: the buffers are rebuilt at the new capacity
        if self._pipe is None:
            self._bits_key = None   # (a new pipeline has no occupancy yet - and may reuse the id() of the one just dropped)
        if self._pipe is None or self._pipe.field.device != device:
            self._bits_key = None
            g, r, vol = self.coarse_geo_net, self.coarse_radiance_net, self.obj_bound.volume
            e = g.embed_fn
            side = float(e.max_xyz[0] - e.min_xyz[0])
            cfg = NgpConfig(n_levels=e.n_levels, n_feat_per_entry=e.n_feat_per_entry, hashmap_size=int(round(torch.log2(torch.tensor(float(e.hashmap_size))).item())),
                            base_res=e.base_res, max_res=e.max_res, side=side,
                            origin=tuple(float(v) for v in (e.max_xyz + e.min_xyz) / 2.0), geo_W=g.W, geo_D=g.D, W_feat=g.W_feat,
                            rad_W=r.W, rad_D=r.D, sh_degree=r.embed_fn_view.n_freqs, rad_mode=r.mode, n_grid=vol.get_n_grid(),
                            n_sample=self.get_n_coarse_sample(), near_distance=self.obj_bound.get_optim_cfgs('near_distance'),
                            add_inf_z=bool(self.add_inf_z), white_bkg=bool(self.get_ray_cfgs('white_bkg')),
                            noise_std=float(self.get_ray_cfgs('noise_std') or 0.0))
            assert abs(vol.get_diag_len() - (3.0 * side * side) ** 0.5) < 1e-5, 'encoder volume and bound volume differ'
            fld = NgpField.__new__(NgpField)  # metadata only: the parameters stay in the nn.Modules
            fld.cfg, fld.device = cfg, device
            fld.resolutions, fld.offsets = e.resolutions, e.offsets
            fld.min_xyz, fld.max_xyz = e.min_xyz.tolist(), e.max_xyz.tolist()
            fld.grid_desc, fld.geo_desc, fld.rad_desc = e.desc, g.layers.desc, r.layers.desc
            fld.geo_dims, fld.rad_dims = g.layers.dims, r.layers.dims
            fld.geo_out_dim, fld.feat_off = g.layers.dims[-1], 0
            fld.n_params = 0
            fld.params = fld.grads = torch.zeros(4, device=device)
            fld._seg = {}
            max_rays = int(self.chunk_rays) if self.chunk_rays and self.chunk_rays > 0 else 32768
            self._pipe = NgpPipeline(fld, max_rays=max_rays, max_samples=max(1 << 20, 2 * max_rays, int(min_samples)), packed_bits=True)
            # the process-wide sampler stream of the native module the bound samples with, like the reference's static generators
            if isinstance(self.obj_bound, BitfieldBound):
                from ..ops.bitfield_func import bitfield_rng
                self._pipe.rng = bitfield_rng()
            else:
                from ..ops.volume_func import sampler_rng
                self._pipe.rng = sampler_rng()
        if isinstance(self.obj_bound, BitfieldBound):
            self._pipe.set_occupancy_bits(self.obj_bound.density_bitfield, 2)  # Morton bits, marched in place
        else:
            # the bool grid is packed to bits for the marcher (seven torch launches over 2 M voxels): only when it has changed - every
            # writer of Volume.bitfield works in place, which moves the tensor's version counter
            bf = self.obj_bound.volume.get_voxel_bitfield(flatten=True)
            key = (id(self._pipe), bf.data_ptr(), bf._version)
            if getattr(self, '_bits_key', None) != key:
                self._pipe.set_bitfield(bf)
                self._bits_key = key
        return self._pipe

    def _sample_packed(self, rays_o, rays_d, exact, defer=False):
        """March the rays into the packed buffers and make sure no sample is dropped SILENTLY: the packed buffers have a fixed
        capacity and the scan clamps the segments to it, while the reference's dense (R, n_sample) tensors hold every sample (first
        steps with an all-ones bitfield, 32768-ray inference chunks).  Only when R * n_sample can exceed the capacity:
          exact (inference): the marcher's total is read back (one host read, where the reference's FgModel.forward has three) and,
            if it does not fit, the buffers are rebuilt 1.25x larger than needed and the SAME launch of the sampler's pcg32 stream is
            repeated - nothing is ever truncated;
          training: a host read per step would serialise the host with the device (measured: 1.04 -> 3.75 ms/step with
            torch.optim.Adam), so the total travels to pinned memory asynchronously and is read at the NEXT call, where it (a) sizes
            the buffers for the coming step - 1.5x the last step's samples per ray must fit, else they grow before marching - and (b)
            reports a step that overflowed all the same (samples per ray more than 1.5x the previous step's) with a warning.  The
            FIRST training step has no history and takes the exact path (one host read): with the all-ones bitfield of a fresh model
            it needs R * n_sample samples, four times the default capacity at 4096 rays, and nothing is dropped."""
        self._check_deferred_overflow(rays_o.device)
        pipe = self._packed_pipeline(rays_o.device)
        R = rays_o.shape[0]
        check = R * pipe.cfg.n_sample > pipe.cap
        if check and not exact:
            # training: no sample may be dropped either (the reference's dense tensors hold them all, fg_model.py:252-262).  The last
            # step's samples-per-ray (read back a step late, no stall) bounds this one: grow BEFORE marching when 1.5x that rate does not
            # fit; without a history (the first step of a run, all-ones bitfield: R * n_sample samples) take the exact path once.
            rate = getattr(self, '_samples_per_ray', None)
            if rate is None:
                exact = True
            elif rate * R * 1.5 > pipe.cap:
                pipe = self._packed_pipeline(rays_o.device, min_samples=min(R * pipe.cfg.n_sample, (int(rate * R * 1.5) + 1023) // 1024 * 1024))
                check = R * pipe.cfg.n_sample > pipe.cap
        state = pipe.rng.state
        pipe.sample(rays_o, rays_d)
        self._exact_pending = None
        if check:
            if exact and defer:
                # exact, but OPTIMISTIC (inference chunks): the scan's total travels to pinned memory while the chunk's render is queued, and
                # _resolve_exact reads it AFTER that - the host never waits for the device between the marcher and the render, which left the
                # GPU idle for the ~10 launches of every chunk (one 800 x 800 view through the module: 15.3 -> 13.6 ms, the tiled scan included).  A total that reads
                # "full" makes _resolve_exact grow the buffers and repeat the chunk from the same generator state: nothing is ever truncated.
                if getattr(self, '_exact_host', None) is None:
                    self._exact_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._exact_host.copy_(pipe.n_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._exact_pending = (ev, state, pipe.cap, R)
            elif exact:
                need = int(pipe.buf['counts'][:R].sum(dtype=torch.int64))
                self._samples_per_ray = need / max(1, R)
                if need > pipe.cap:
                    pipe = self._packed_pipeline(rays_o.device, min_samples=(need * 5 // 4 + 1023) // 1024 * 1024)
                    pipe.rng.set_state(state)
                    pipe.sample(rays_o, rays_d)
                    assert int(pipe.n_dev.item()) == need
            else:
                # the scan's total (offsets[R]) is exact unless it was clamped to the capacity: one asynchronous copy of that word per step;
                # the per-ray counts are only summed when it reads "full" (then: with them, the number the buffers must grow to)
                if getattr(self, '_ovf_host', None) is None:
                    self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._ovf_host.copy_(pipe.n_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending_ovf = (ev, pipe.cap, R, pipe.buf['counts'])
        return pipe

    def _resolve_exact(self, rays_o, rays_d):
        """the second half of _sample_packed(exact, defer): None when the chunk's samples fitted (the render that was queued meanwhile stands),
        else the pipeline rebuilt at the needed capacity with the SAME samples marched again (the caller renders the chunk again)"""
        pend, self._exact_pending = getattr(self, '_exact_pending', None), None
        if pend is None:
            return None
        ev, state, cap, R = pend
        ev.synchronize()
        total = int(self._exact_host[0])
        if total < cap:      # (the scan clamps at the capacity: a total below it is exact)
            self._samples_per_ray = total / max(1, R)
            return None
        pipe = self._packed_pipeline(rays_o.device)
        need = int(pipe.buf['counts'][:R].sum(dtype=torch.int64))
        self._samples_per_ray = need / max(1, R)
        if need <= pipe.cap:
            return None
        pipe = self._packed_pipeline(rays_o.device, min_samples=(need * 5 // 4 + 1023) // 1024 * 1024)
        pipe.rng.set_state(state)
        pipe.sample(rays_o, rays_d)
        assert int(pipe.n_dev.item()) == need
        return pipe

    def _check_deferred_overflow(self, device):
        pend = getattr(self, '_pending_ovf', None)
        if pend is None:
            return
        self._pending_ovf = None
        ev, cap, R, counts = pend
        ev.synchronize()    # recorded a whole step ago: no wait in practice
        need = int(self._ovf_host[0])
        if need >= cap and counts is not None:
            # the scan clamped: the exact demand from the per-ray counts (still those of that step unless this model has sampled since -
            # then the sum is the newer step's demand, just as good a number to grow to)
            need = max(need, int(counts[:R].sum(dtype=torch.int64)))
        self._samples_per_ray = need / max(1, R)
        if need > cap:
            import warnings
            warnings.warn('packed NGP path: the previous training step asked for {} samples for {} rays but the buffers hold {}; the '
                          'rays behind the fill point were left out of that step (background colour, no gradient: never a ray rendered from a '
                          'truncated sample set).  Growing the buffers to '
                          '{} samples now (set model.chunk_rays lower, or render with inference_only=True for the exact '
                          'path).'.format(need, R, cap, (need * 5 // 4 + 1023) // 1024 * 1024))
            self._packed_pipeline(device, min_samples=(need * 5 // 4 + 1023) // 1024 * 1024)

    def train(self, mode=True):
        """leaving training mode (evaluation, the end of a run) also reports an overflow of the LAST training step"""
        if not mode and getattr(self, '_pending_ovf', None) is not None:
            self._check_deferred_overflow(next(self.parameters()).device)
        return super().train(mode)

    def _forward_packed(self, inputs, inference_only):
        rays_o, rays_d, bkg = inputs['rays_o'].contiguous().float(), inputs['rays_d'].contiguous().float(), inputs['bkg_color']
        train = torch.is_grad_enabled() and not inference_only
        pipe = self._sample_packed(rays_o, rays_d, exact=not train, defer=not train)
        noise_std = float(self.get_ray_cfgs('noise_std') or 0.0) if not inference_only else 0.0
        g, r = self.coarse_geo_net, self.coarse_radiance_net
        rgb, depth, mask, counts = _PackedRenderFn.apply(rays_o, rays_d, bkg, g.embed_fn.embeddings, g.layers.params,
                                                        r.layers.params, pipe, train, noise_std)
        if not train:
            again = self._resolve_exact(rays_o, rays_d)
            if again is not None:      # the buffers were too small for this chunk: rendered again from the same samples, complete
                pipe = again
                rgb, depth, mask, counts = _PackedRenderFn.apply(rays_o, rays_d, bkg, g.embed_fn.embeddings, g.layers.params,
                                                                r.layers.params, pipe, train, noise_std)
        if not inference_only:
            self.adjust_dynamicbs_factor(n_valid=pipe.n_dev[0])
        rgb, depth = self._packed_defaults(rgb, depth, counts, bkg)
        out = {'rgb': rgb, 'depth': depth, 'mask': mask}
        if inference_only:
            return out
        return {k + '_coarse': v for k, v in out.items()}

    def _packed_defaults(self, rgb, depth, counts, bkg):
        """defaults of FgModel.update_values_for_invalid_rays for rays without samples"""
        hit = counts > 0
        far = getattr(self, '_depth_far_dev', None)    # (a one-element tensor kept on the device: no fill launch per call)
        if far is None or far.device != depth.device or float(self.render_cfgs['depth_far']) != self._depth_far_val:
            self._depth_far_val = float(self.render_cfgs['depth_far'])
            far = self._depth_far_dev = torch.full((1,), self._depth_far_val, dtype=depth.dtype, device=depth.device)
        depth = torch.where(hit, depth, far)
        if bkg is None:
            dflt = torch.tensor(self.render_cfgs['bkg_color'], dtype=rgb.dtype, device=rgb.device)[None]
            rgb = torch.where(hit[:, None], rgb, dflt.expand_as(rgb))
        return rgb, depth

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        if not get_progress and self.packed_path_eligible() and inputs['rays_o'].is_cuda:
            return self._forward_packed(inputs, inference_only)
        return super().forward(inputs, inference_only, get_progress, cur_epoch, total_epoch)

    def surface_render(self, inputs, method='secant_root_finding', n_step=128, n_iter=20, threshold=0.01, level=50.0, grad_dir='descent'):
        """a density has no exact surface: the `level` crossing found by the secant search only (nerf_model.py:119-135)"""
        assert grad_dir == 'descent', 'Invalid for density model in nerf...'
        assert method != 'sphere_tracing', 'Do not support for density model in nerf...'
        return super().surface_render(inputs, method, n_step, n_iter, threshold, level, grad_dir)

    # ---- dense reference-shaped path ----------------------------------------------------------------------
    def _forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d, zvals = inputs['rays_o'], inputs['rays_d'], inputs['zvals']
        mask_pts, bkg_color = inputs['mask_pts'], inputs['bkg_color']
        output = {}
        sigma, radiance = self.get_sigma_radiance_by_mask_pts(self.coarse_geo_net, self.coarse_radiance_net, rays_o, rays_d, zvals,
                                                              mask_pts, inference_only)
        out_c = self.ray_marching(sigma, radiance, zvals, inference_only=inference_only, bkg_color=bkg_color)
        weights_c = out_c['weights']
        output['coarse'] = self.output_get_progress(out_c, get_progress)
        if self.get_ray_cfgs('n_importance') > 0:
            zvals, mask_pts = self.upsample_zvals(zvals, weights_c, mask_pts, inference_only)
            sigma, radiance = self.get_sigma_radiance_by_mask_pts(self.fine_geo_net, self.fine_radiance_net, rays_o, rays_d, zvals,
                                                                  mask_pts, inference_only)
            out_f = self.ray_marching(sigma, radiance, zvals, inference_only=inference_only, bkg_color=bkg_color)
            output['fine'] = self.output_get_progress(out_f, get_progress)
        return self.adjust_coarse_fine_output(output, inference_only)

    def upsample_zvals(self, zvals, weights, mask_pts=None, inference_only=True):
        """coarse weights[1:n-1] on the interval mid-points -> n_importance inverse-CDF samples, merged and sorted
        (nerf_model.py:93-117)"""
        w = weights[:, 1:self.get_ray_cfgs('n_sample') - 1]
        mids = 0.5 * (zvals[..., 1:] + zvals[..., :-1])
        det = True if inference_only else (not self.get_ray_cfgs('perturb'))
        new = sample_pdf(mids.contiguous(), w.detach().contiguous(), self.get_ray_cfgs('n_importance'), det).detach()
        zvals, _ = torch.sort(torch.cat([zvals, new], -1), -1)
        return zvals, self.merge_full_mask(mask_pts, new)
