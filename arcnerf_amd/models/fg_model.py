"""FgModel (arcnerf/models/fg_model.py:19-470): foreground model bounded by an object structure.  Rays that miss the
bound (or carry no valid sample) are skipped and filled with defaults; the valid samples of the remaining rays are
compacted before the networks run and scattered back with the "repeat the last valid sample" padding.

This is the DENSE, reference-shaped path (it keeps the reference's host decisions such as `reduce_empty_mask`); the
instant-ngp configuration takes the packed, sync-free path of NeRF._forward_packed instead.
"""
import torch

from ..geometry.ray import get_ray_points_by_zvals, normalize
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.torch_utils import chunk_processing
from .base_3d_model import Base3dModel
from .surface_render import render_surface
from .base_modules.obj_bound import build_obj_bound


class FgModel(Base3dModel):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        self.obj_bound, self.obj_bound_type = build_obj_bound(cfgs.model)
        self.render_cfgs = self.read_render_cfgs()

    def read_render_cfgs(self):
        p = {}
        ob = get_value_from_cfgs_field(self.cfgs.model, 'obj_bound')
        if ob is None:
            p.update(bkg_color=[1.0, 1.0, 1.0], depth_far=10.0, normal=[0.0, 1.0, 0.0], max_allowance=-1)
        else:
            p['bkg_color'] = get_value_from_cfgs_field(ob, 'bkg_color', [0.0, 0.0, 0.0])
            p['depth_far'] = get_value_from_cfgs_field(ob, 'depth_far', 10.0)
            p['normal'] = get_value_from_cfgs_field(ob, 'normal', [0.0, 1.0, 0.0])
            p['max_allowance'] = get_value_from_cfgs_field(ob, 'log_max_allowance', -1)
            if p['max_allowance'] > 0:
                p['max_allowance'] = 1 << p['max_allowance']
        p['measured_batch_size'] = 0
        p['measured_count'] = 0
        return p

    def get_render_cfgs(self, key=None):
        return self.render_cfgs if key is None else self.render_cfgs[key]

    def set_render_cfgs(self, key, value):
        self.render_cfgs[key] = value

    def get_n_coarse_sample(self):
        return self.get_ray_cfgs('n_sample')

    def get_obj_bound(self):
        return self.obj_bound

    def get_obj_bound_type(self):
        return self.obj_bound_type

    def get_obj_bound_structure(self):
        return self.obj_bound.get_obj_bound()

    def get_optim_cfgs(self, key=None):
        return self.obj_bound.get_optim_cfgs(key)

    def set_optim_cfgs(self, key, value):
        return self.obj_bound.set_optim_cfgs(key, value)

    # ---- dynamic batch size (fg_model.py:100-130): factor = mean over steps of max_allowance / (valid samples + 1) ----
    # (trainer.DynamicBsMeter: the step's count is one device copy into a ring, the reference's double arithmetic runs when the factor
    # is read; render_cfgs['measured_count'] / ['measured_batch_size'] stay readable like the reference's)
    def _meter(self):
        m = getattr(self, '_dynbs_meter', None)
        if m is None or m.max_allowance != self.render_cfgs['max_allowance']:
            from ..trainer.dynamic_bs import DynamicBsMeter
            m = self._dynbs_meter = DynamicBsMeter(self.render_cfgs['max_allowance'])
        return m

    def _sync_measurement(self):
        m = self._meter()
        self.render_cfgs['measured_batch_size'], self.render_cfgs['measured_count'] = m.measured_batch_size, m.measured_count

    def reset_measurement(self):
        self._meter().reset()
        self._sync_measurement()

    def adjust_dynamicbs_factor(self, mask_pts=None, n_valid=None, stream=None):
        if self.render_cfgs['max_allowance'] <= 0 or (mask_pts is None and n_valid is None):
            return
        self._meter().add(mask_pts.sum() if n_valid is None else n_valid, stream=stream)
        self.render_cfgs['measured_count'] = self._meter().measured_count

    def get_dynamicbs_factor(self):
        f = self._meter().factor()
        self._sync_measurement()
        return f

    # ---- bounds / samples ----------------------------------------------------------------------------
    @torch.no_grad()
    def get_near_far_from_rays(self, inputs):
        return self.obj_bound.get_near_far_from_rays(inputs, near_hardcode=self.get_ray_cfgs('near'),
                                                     far_hardcode=self.get_ray_cfgs('far'),
                                                     bounding_radius=self.get_ray_cfgs('bounding_radius'))

    @torch.no_grad()
    def get_zvals_from_near_far(self, near, far, n_pts, inference_only=False, rays_o=None, rays_d=None):
        return self.obj_bound.get_zvals_from_near_far(near, far, n_pts, inference_only, self.get_ray_cfgs('inverse_linear'),
                                                      self.get_ray_cfgs('perturb'), rays_o=rays_o, rays_d=rays_d)

    @torch.no_grad()
    def reduce_empty_mask(self, zvals, mask_pts):
        if mask_pts is None:
            return zvals, mask_pts
        keep = max(2, int(mask_pts.sum(dim=1).max()))
        return zvals[:, :keep], mask_pts[:, :keep]

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d, bkg_color = inputs['rays_o'], inputs['rays_d'], inputs['bkg_color']
        with torch.no_grad():
            near, far, mask_rays = self.get_near_far_from_rays(inputs)
            zvals, mask_pts = self.get_zvals_from_near_far(near, far, self.get_n_coarse_sample(), inference_only, rays_o, rays_d)
            zvals, mask_pts = self.reduce_empty_mask(zvals, mask_pts)
        inputs['zvals'], inputs['mask_pts'] = zvals, mask_pts
        if mask_rays is None:
            if mask_pts is not None:
                raise RuntimeError('This case should not happen...Check it')
            return self._forward(inputs, inference_only, get_progress, cur_epoch, total_epoch)
        if mask_pts is not None:
            mask_rays = torch.logical_and(mask_rays, torch.any(mask_pts, dim=1))
        if bool(torch.all(mask_rays)):
            return self._forward(inputs, inference_only, get_progress, cur_epoch, total_epoch)
        # ONE nonzero (one host sync) for the whole split: every `v[mask]` / `full[mask] = v` on the same mask would repeat it
        idx_rays = mask_rays.nonzero(as_tuple=True)[0]
        z_valid = zvals.index_select(0, idx_rays)
        m_valid = mask_pts.index_select(0, idx_rays) if mask_pts is not None else None
        empty = idx_rays.numel() == 0
        if empty:  # all-background batch: one synthetic ray gives the output keys, then everything takes the defaults
            mask_rays = mask_rays.clone()
            mask_rays[0] = True
            idx_rays = torch.zeros(1, dtype=torch.long, device=zvals.device)
            z_valid = torch.zeros((1, zvals.shape[1]), dtype=zvals.dtype, device=zvals.device)
            z_valid[0, 1:] = 1.0
            if mask_pts is not None:
                m_valid = torch.zeros((1, mask_pts.shape[1]), dtype=torch.bool, device=zvals.device)
                m_valid[0, :2] = True
        sub = {k: (v.index_select(0, idx_rays) if isinstance(v, torch.Tensor) else v) for k, v in inputs.items()}
        sub['zvals'], sub['mask_pts'] = z_valid, m_valid
        sub['bkg_color'] = bkg_color.index_select(0, idx_rays) if bkg_color is not None else None
        out_valid = self._forward(sub, inference_only, get_progress, cur_epoch, total_epoch)
        if empty:
            mask_rays[0] = False
            idx_rays = idx_rays[:0]
            out_valid = {k: (v[:0] if isinstance(v, torch.Tensor) else v) for k, v in out_valid.items()}
        return self.update_values_for_invalid_rays(out_valid, mask_rays, bkg_color, idx_rays)

    def _forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        raise NotImplementedError('implement _forward (rays with coarse zvals) in the child class')

    # ---- networks on valid samples only -------------------------------------------------------------------
    def get_sigma_radiance_by_mask_pts(self, geo_net, radiance_net, rays_o, rays_d, zvals, mask_pts=None, inference_only=False):
        n_rays, n_pts = zvals.shape
        pts = get_ray_points_by_zvals(rays_o, rays_d, zvals)
        dirs = rays_d.unsqueeze(1).expand(n_rays, n_pts, 3)
        if mask_pts is None:
            pts, dirs = pts.reshape(-1, 3), dirs.reshape(-1, 3)
        else:
            flat = mask_pts.reshape(-1).nonzero(as_tuple=True)[0]   # one nonzero for the two gathers and the two scatters
            pts = pts.reshape(-1, 3).index_select(0, flat)
            dirs = rays_d.index_select(0, torch.div(flat, n_pts, rounding_mode='floor'))   # the expanded view is never materialised
            if not inference_only:
                self.adjust_dynamicbs_factor(mask_pts)
        _sigma, _radiance = self.field_on_points(geo_net, radiance_net, pts.contiguous(), dirs.contiguous())
        if mask_pts is None:
            return _sigma.view(n_rays, -1), _radiance.view(n_rays, -1, 3)
        last = torch.cumsum(mask_pts.sum(dim=1), dim=0) - 1
        sigma = _sigma[last].unsqueeze(1).repeat(1, n_pts)
        radiance = _radiance[last].unsqueeze(1).repeat(1, n_pts, 1)
        sigma = sigma.view(-1).index_copy(0, flat, _sigma.reshape(-1)).view(n_rays, n_pts)
        radiance = radiance.view(-1, 3).index_copy(0, flat, _radiance.reshape(-1, 3)).view(n_rays, n_pts, 3)
        return sigma, radiance

    def update_values_for_invalid_rays(self, output_valid, mask, rand_bkg_color=None, idx=None):
        """defaults for rays that never entered _forward: rgb* = bkg colour, depth* = depth_far, mask* = 0, normal* = cfg
        normal, progress_trans_shift* = 1, progress_sigma* = -1 for sdf models, other progress_* = 0 (fg_model.py:320-387)"""
        n_rays = mask.shape[0]
        out = {}
        for k, v in output_valid.items():
            if not isinstance(v, torch.Tensor):
                out[k] = v
                continue
            shape = (n_rays, *v.shape[1:])
            kw = dict(dtype=v.dtype, device=v.device)
            if k.startswith('rgb'):
                if rand_bkg_color is not None:
                    full = torch.ones(shape, **kw) * rand_bkg_color
                else:
                    full = torch.ones(shape, **kw) * torch.tensor(self.render_cfgs['bkg_color'], **kw)[None]
            elif k.startswith('depth'):
                full = torch.full(shape, float(self.render_cfgs['depth_far']), **kw)
            elif k.startswith('mask'):
                full = torch.zeros(shape, **kw)
            elif k.startswith('normal'):
                nrm = normalize(torch.tensor(self.render_cfgs['normal'], **kw)[None])
                full = torch.ones(shape, **kw) * (nrm[None] if k == 'normal_pts' else nrm)
            elif k.startswith('progress'):
                if 'sigma' in k and self.sigma_reverse():
                    full = -torch.ones(shape, **kw)
                elif 'trans_shift' in k:
                    full = torch.ones(shape, **kw)
                else:
                    full = torch.zeros(shape, **kw)
            else:
                out[k] = v
                continue
            if idx is None:
                idx = mask.nonzero(as_tuple=True)[0]
            out[k] = full.index_copy(0, idx, v)
        return out

    @staticmethod
    def merge_full_mask(mask_pts, zvals_new):
        if mask_pts is None:
            return None
        new = torch.ones_like(zvals_new, dtype=torch.uint8)
        return torch.sort(torch.cat([mask_pts.type(torch.uint8), new], -1), -1, descending=True)[0].type(torch.bool)

    def optimize(self, cur_epoch=0):
        self.obj_bound.optimize(cur_epoch, self.get_n_coarse_sample(), self.get_est_opacity)

    def surface_render(self, inputs, method='sphere_tracing', n_step=128, n_iter=100, threshold=0.01, level=50.0, grad_dir='descent'):
        """inference only: colour of the first surface point of every ray; rays that miss the object bound are skipped
        (fg_model.py:412-470, base_3d_model.py:307-366)"""
        return render_surface(self, inputs, method, n_step, n_iter, threshold, level, grad_dir)
