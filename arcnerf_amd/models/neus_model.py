"""NeuS (arcnerf/models/neus_model.py): the model class (:21-218) and its per-interval math sdf_to_cdf / sdf_to_pdf /
sdf_to_alpha (:221-265).  SURVEY.md section 8f, rank 1.

What runs where: the sphere bound, sdf_to_alpha (fwd + bwd), the inverse-CDF up-sampling and the compositing (alpha= branch)
are HIP kernels; the geometry / radiance nets of configs/models/neus.yaml are 256-wide linear stacks on the MFMA products of
csrc/gemm.hip (ops.autograd.linear: closed under differentiation), whose input gradient - the normal - is taken by autograd with
create_graph=True, so the Eikonal term and everything downstream of the normals differentiate a second time.  The hash-grid + fused-MLP variant
(NeuS-NGP) needs a second-order backward of those kernels and is not built yet."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..geometry.ray import get_ray_points_by_zvals, normalize
from ..ops import functional as Fn
from ..ops.autograd import NeusPackedRenderFn, NeusSlotsFn, SdfToAlphaFn
from ..render.ray_helper import alpha_to_weights, sample_pdf
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.registry import MODEL_REGISTRY
from ..utils.torch_utils import chunk_processing
from .base_modules import build_geo_model, build_radiance_model
from .masked_samples import hold_rays
from .sdf_model import SdfModel


@MODEL_REGISTRY.register()
class Neus(SdfModel):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        self.geo_net = build_geo_model(self.cfgs.model.geometry)
        self.radiance_net = build_radiance_model(self.cfgs.model.radiance)
        self.ray_cfgs['n_importance'] = get_value_from_cfgs_field(self.cfgs.model.rays, 'n_importance', 0)
        self.ray_cfgs['n_iter'] = get_value_from_cfgs_field(self.cfgs.model.rays, 'n_iter', 4)
        self.radius_init = get_value_from_cfgs_field(self.cfgs.model.geometry, 'radius_init', 1.0)
        self.inv_s, self.speed_factor = self.get_params()
        self.anneal_end = get_value_from_cfgs_field(self.cfgs.model.params, 'anneal_end', 0)
        self.radius_bound = get_value_from_cfgs_field(self.cfgs.model.rays, 'radius_bound', 1.5)

    def get_net(self):
        return self.geo_net, self.radiance_net

    # ---- packed path: a pruned foreground without the padded (rays, P) tensors ------------------------------------------------------
    use_packed_path = True    # set False to force the dense reference-shaped path (tests compare the two)

    def packed_path_eligible(self):
        """occupancy-marched foreground (VolumeBound with the sparse sampler), no importance sampling (the marcher's samples are the
        samples: `n_importance` 0 as in capture_qqtiger_neusngp_multivol.yaml), no hard-coded near / far, no white background"""
        from .base_modules.obj_bound import VolumeBound
        b = self.obj_bound
        return (self.use_packed_path and isinstance(b, VolumeBound) and b.uses_sparse_sampling() and self.get_ray_cfgs('n_importance') <= 0
                and self.get_ray_cfgs('near') is None and self.get_ray_cfgs('far') is None and not self.get_ray_cfgs('white_bkg')
                and not self.get_ray_cfgs('bounding_radius'))

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        """get_progress may be the string 't_last' (FullModel's rgb blending needs only the foreground's last transmittance,
        full_model.py:278-330): the packed path then returns `progress_trans_shift` as (rays, 1) and nothing per sample"""
        if (get_progress is False or get_progress == 't_last') and inputs['rays_o'].is_cuda and self.packed_path_eligible():
            return self._forward_packed(inputs, inference_only, get_progress == 't_last', cur_epoch)
        return super().forward(inputs, inference_only, bool(get_progress), cur_epoch, total_epoch)

    @torch.no_grad()
    def _sample_begin(self, rays_o, rays_d):
        """bounds + occupancy marcher + the scans of the section layout queued, the totals on their way to the host (no wait here)"""
        from ..ops.volume_func import sampler_rng
        vol, n_pts = self.obj_bound.volume, self.get_n_coarse_sample()
        rng = sampler_rng()
        zd, counts, _, _ = Fn.march_count(rays_o, rays_d, vol.get_range23(), vol.get_n_grid(),
                                          vol.get_voxel_bitfield(), n_pts, vol.get_diag_len() / n_pts,
                                          self.obj_bound.get_optim_cfgs('near_distance'), rng.state, rng.inc)
        rng.advance()
        h = Fn.neus_pack_begin(zd, counts)
        hold_rays(h, rays_o, rays_d)        # (keeps the rays alive while a later forward may still match this handle by their pointers)
        return h

    def presample(self, inputs):
        """March the rays of a LATER forward now (FullModel.prefetch_samples calls this on the sampling stream while the current step's
        backward runs): that forward, on the same ray tensors, picks the samples up and reads totals that arrived long ago - no marcher
        and no host wait on its critical path.  One sampler launch per batch in batch order, as without it: the same samples."""
        rays_o, rays_d = inputs['rays_o'], inputs['rays_d']
        if not (self.packed_path_eligible() and rays_o.is_cuda and rays_o.is_contiguous() and rays_d.is_contiguous()
                and rays_o.dtype == torch.float32 and rays_d.dtype == torch.float32):
            return
        self._presampled = ((rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], rays_o._version, rays_d._version),
                            self._sample_begin(rays_o, rays_d))

    def _forward_packed(self, inputs, inference_only, want_t_last, cur_epoch):
        """March (K2 + K3 in one launch) -> section layout of the marched samples (arcn_neus_sections) -> the nets on the packed mid
        points (the same autograd graph as the dense path: hash encoder, sdf net with its input Jacobian, radiance net) -> ONE render
        kernel per direction (slope, cos annealing, sdf_to_alpha, weights, sums, invalid-ray defaults).  One host read (the number of
        points and the longest ray) where the dense path has five; no boolean-mask gathers, no padded scatters, no (rays, P) tensor
        except the `normal_pts` output the Eikonal loss consumes (a gather of the packed normals, padded slots included)."""
        rays_o, rays_d = inputs['rays_o'].contiguous().float(), inputs['rays_d'].contiguous().float()
        bkg_color = inputs['bkg_color']
        n_rays = rays_o.shape[0]
        train = not inference_only
        with torch.no_grad():
            pre, self._presampled = getattr(self, '_presampled', None), None
            if pre is not None and pre[0] == (rays_o.data_ptr(), rays_d.data_ptr(), n_rays, rays_o._version, rays_d._version):
                handle = pre[1]
                torch.cuda.current_stream().wait_event(handle['event'])       # marched on the sampling stream while the previous step ran
                for t_ in (handle['z'], handle['counts'], handle['offsets'], handle['kmax'], handle['n_eval']) + handle['keep']:
                    t_.record_stream(torch.cuda.current_stream())
            else:
                handle = self._sample_begin(rays_o, rays_d)
            pk = Fn.neus_pack_end(handle, float(self.get_ray_cfgs('n_sample')))
            total = pk['total']
            if total > 0:
                pts, dirs = Fn.packed_points(rays_o, rays_d, pk['t_mid'], pk['ray_id'])
        dflt_rgb = self.render_cfgs['bkg_color']
        nv = torch.tensor(self.render_cfgs['normal'], dtype=torch.float32)
        dflt_nrm = (nv / (nv.norm() + 1e-8)).tolist()
        if total > 0:
            if train:
                self.adjust_dynamicbs_factor(n_valid=pk['offsets'][n_rays])
            sdf, radiance, normal = self.field_with_normal(self.geo_net, self.radiance_net, pts, dirs)
        else:   # nothing marched anywhere: every ray takes the defaults
            sdf = rays_o.new_zeros((1,))
            radiance = normal = rays_o.new_zeros((1, 3))
        cos_anneal = 1.0 if inference_only else self.get_cos_anneal(cur_epoch)
        scale = self.forward_scale()
        rgb, depth, mask, nrm, t_last = NeusPackedRenderFn.apply(sdf, radiance, normal, scale, pk, rays_d, cos_anneal,
                                                                 bkg_color.contiguous().float() if bkg_color is not None else None,
                                                                 float(self.render_cfgs['depth_far']), dflt_rgb, dflt_nrm)
        out = {'rgb': rgb, 'depth': depth, 'mask': mask, 'normal': nrm}
        if train:
            # (the reference hands out a Python float, neus_model.py:100-104 - a host read per step; the packed path keeps the 0-d device
            # tensor: float(), format() and logging work on it, and nothing waits for the device here)
            out['params'] = {'scale': scale.detach().reshape(())}
            # the dense (rays, P, 3) per-slot normals of the reference's output (padded slots repeat the ray's last point, rays without
            # samples hold the default normal): one kernel each way (an index_select would send the gradients of every padded slot
            # through atomics on one row)
            out['normal_pts'] = NeusSlotsFn.apply(normal, pk['offsets'], pk['p_dense'], dflt_nrm)
        if want_t_last:
            out['progress_trans_shift'] = t_last[:, None]
        return out

    def get_params(self):
        """inv_s = -log(init_var) / speed_factor, learnable (neus_model.py:45-53)"""
        dtype = next(self.parameters()).dtype
        init_var = get_value_from_cfgs_field(self.cfgs.model.params, 'init_var', 0.05)
        speed_factor = get_value_from_cfgs_field(self.cfgs.model.params, 'speed_factor', 10)
        return nn.Parameter(torch.tensor([-np.log(init_var) / speed_factor], dtype=dtype), requires_grad=True), speed_factor

    def forward_scale(self):
        """scale = exp(inv_s * speed)"""
        return torch.exp(self.inv_s * self.speed_factor)

    def get_cos_anneal(self, cur_epoch):
        return 1.0 if self.anneal_end == 0 else min(1.0, cur_epoch / self.anneal_end)

    def _forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d, zvals = inputs['rays_o'], inputs['rays_d'], inputs['zvals']
        mask_pts, bkg_color = inputs['mask_pts'], inputs['bkg_color']
        zvals, mask_pts = self.upsample_zvals(rays_o, rays_d, zvals, mask_pts, inference_only)
        mid_zvals, zvals, mask_mid_pts = self.handle_mid_pts(zvals, mask_pts)
        sdf, radiance, normal_pts = self.get_sdf_radiance_normal_by_mask_pts(self.geo_net, self.radiance_net, rays_o, rays_d,
                                                                             mid_zvals, mask_mid_pts, inference_only)
        rays_d_repeat = torch.repeat_interleave(rays_d.unsqueeze(1), mid_zvals.shape[1], dim=1)
        cos_anneal_ratio = 1.0 if inference_only else self.get_cos_anneal(cur_epoch)
        # rays and normals are opposite: the slope (their dot product) is negative
        slope = torch.sum(rays_d_repeat * normal_pts, dim=-1, keepdim=True)[..., 0]
        iter_slope = -(F.relu(-slope * 0.5 + 0.5) * (1 - cos_anneal_ratio) + F.relu(-slope) * cos_anneal_ratio)
        alpha = sdf_to_alpha(sdf, zvals, iter_slope, self.forward_scale())
        output = self.ray_marching(sdf, radiance, mid_zvals, alpha=alpha, inference_only=inference_only, bkg_color=bkg_color)
        output['normal'] = torch.sum(output['weights'].unsqueeze(-1) * normalize(normal_pts), -2)
        if not inference_only:
            output['params'] = {'scale': self.forward_scale().detach().reshape(())}   # (0-d device tensor: no host read per step)
            output['normal_pts'] = normal_pts
        return self.output_get_progress(output, get_progress)

    def _crossing_weights(self, zvals, sdf, dist_to_origin, sharpness):
        """compositing weights of the sections under a fixed sigmoid sharpness: large where the sdf crosses zero.  The section
        slope is the finite difference, replaced by the previous section's when that is steeper-downwards-limited (the min of
        the two, in [-10, 0]) and zeroed for sections entirely outside the radius bound (neus_model.py:126-160)"""
        step = zvals[:, 1:] - zvals[:, :-1]
        slope = (sdf[:, 1:] - sdf[:, :-1]) / (step + 1e-5)
        slope = torch.minimum(F.pad(slope[:, :-1], (1, 0)), slope).clamp(-10.0, 0.0)
        touches_bound = torch.minimum(dist_to_origin[:, :-1], dist_to_origin[:, 1:]) < self.radius_bound
        alpha = sdf_to_alpha((sdf[:, :-1] + sdf[:, 1:]) * 0.5, zvals, slope * touches_bound, sharpness, clip=False)
        return alpha_to_weights(alpha)[1]

    def upsample_zvals(self, rays_o, rays_d, zvals, mask_pts=None, inference_only=False, s=32):
        """n_iter rounds of importance sampling around the sdf zero crossing, the sharpness doubling every round
        (neus_model.py:106-172)"""
        n_new, rounds = self.get_ray_cfgs('n_importance'), self.get_ray_cfgs('n_iter')
        if n_new <= 0:
            return zvals, mask_pts
        deterministic = inference_only or not self.get_ray_cfgs('perturb')
        for rnd in range(rounds):
            # (the new depths are detached from the net - the reference detaches `weights` - so the sdf evaluations of these rounds need
            # no graph: no saved activations, and the dense layers take their activation-in-the-epilogue form)
            with torch.no_grad():
                pts = get_ray_points_by_zvals(rays_o, rays_d, zvals)
                sdf = self.forward_pts(pts.view(-1, 3)).view(zvals.shape)
                weights = self._crossing_weights(zvals, sdf, torch.norm(pts, dim=-1), s * 2 ** (rnd + 1))
            fresh = sample_pdf(zvals.contiguous(), weights.detach().contiguous(), n_new // rounds, deterministic).detach()
            zvals = torch.sort(torch.cat([zvals, fresh], dim=-1), dim=-1)[0]
            mask_pts = self.merge_full_mask(mask_pts, fresh)
        return zvals, mask_pts

    def handle_mid_pts(self, zvals, mask_pts):
        """-> section mid points (B, P), section ends (B, P+1), mask of the mid points: the P samples bound P-1 sections and one
        more is appended behind the last sample (half a coarse step long, a full one in the masked layout) so that every sample
        owns a section (neus_model.py:174-202)"""
        half_step = (zvals[:, -1] - zvals[:, 0]) / self.get_ray_cfgs('n_sample') * 0.5
        if mask_pts is None:
            inner = 0.5 * (zvals[..., 1:] + zvals[..., :-1])
            mid = torch.cat([inner, (inner[:, -1] + half_step)[:, None]], dim=-1)
            return mid, torch.cat([zvals, (zvals[:, -1] + half_step)[:, None]], dim=-1), None
        n_rays, n_pts = zvals.shape
        col = torch.zeros((n_rays, 1), dtype=torch.bool, device=zvals.device)
        beyond = (zvals[:, -1] + half_step * 2.0)[:, None]                         # padded slots and the extra end: beyond the ray
        ends = torch.cat([torch.where(mask_pts, zvals, beyond.expand(n_rays, n_pts)), beyond], dim=1)
        return 0.5 * (ends[..., 1:] + ends[..., :-1]), ends, torch.cat([~col, mask_pts[:, :-1]], dim=1)

    def get_est_opacity(self, dt, pts):
        """opacity of a voxel-sized step towards the origin (neus_model.py:204-218)"""
        rays_d = -normalize(pts)
        sdf, _, normal = chunk_processing(self.geo_net.forward_with_grad, self.chunk_pts, False, pts)
        slope = torch.sum(rays_d * normal, dim=-1, keepdim=True)
        zvals = torch.zeros((pts.shape[0], 2), dtype=pts.dtype, device=pts.device)
        zvals[:, 1] += dt * 1.0 / math.sqrt(3)
        return sdf_to_alpha(sdf, zvals, -F.relu(-slope), self.forward_scale())[:, 0]


def sdf_to_cdf(sdf, s):
    """sigmoid(sdf * s)  (neus_model.py:221-228)"""
    return torch.sigmoid(sdf * s)


def sdf_to_pdf(sdf, s):
    """s e^{-s sdf} / (1 + e^{-s sdf})^2  (neus_model.py:231-239)"""
    esx = torch.exp(-sdf * s)
    return s * esx / ((1 + esx) ** 2)


def sdf_to_alpha(mid_sdf, zvals, mid_slope, s, clip=True):
    """mid_sdf, mid_slope (B, N_pts-1), zvals (B, N_pts), s float or scalar tensor -> alpha (B, N_pts-1)
    (neus_model.py:242-265); differentiable w.r.t. mid_sdf, mid_slope and a tensor s."""
    return SdfToAlphaFn.apply(mid_sdf.contiguous(), zvals.contiguous(), mid_slope.contiguous(), s, clip)
