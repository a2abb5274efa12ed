"""NeRF++ background (arcnerf/models/nerfpp_bkg_model.py:14-136): the scene outside the bounding sphere on inverted-sphere
coordinates (x/r, y/r, z/r, 1/r), sampled on a multi-sphere set of shells (ray / sphere kernel), a coarse and an optional fine
stage like NeRF.  SURVEY.md section 8f, rank 2."""
import torch

from ..geometry.ray import get_ray_points_by_zvals
from ..render.ray_helper import sample_pdf
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.registry import MODEL_REGISTRY
from ..utils.torch_utils import chunk_processing
from .base_modules import build_geo_model, build_radiance_model
from .bkg_model import BkgModel


@MODEL_REGISTRY.register()
class NeRFPP(BkgModel):
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
        assert self.get_ray_cfgs('bounding_radius') is not None, 'Please specify the bounding radius for nerf++ model'

    def get_net(self):
        if self.get_ray_cfgs('n_importance') > 0:
            return self.fine_geo_net, self.fine_radiance_net
        return self.coarse_geo_net, self.coarse_radiance_net

    def _stage(self, geo_net, radiance_net, rays_o, rays_d, zvals, radius, n_pts, inference_only):
        pts = get_ray_points_by_zvals(rays_o, rays_d, zvals)
        if radius is None:  # up-sampled points: their own distance to the origin
            radius = torch.norm(pts, dim=-1)[..., None]
        pts = torch.cat([pts / radius, 1 / radius], dim=-1).view(-1, 4)
        dirs = torch.repeat_interleave(rays_d, n_pts, dim=0)
        sigma, radiance = self.field_on_points(geo_net, radiance_net, pts.contiguous(), dirs.contiguous())
        return self.ray_marching(sigma.view(-1, n_pts), radiance.view(-1, n_pts, 3), zvals, inference_only=inference_only)

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        rays_o, rays_d = inputs['rays_o'], inputs['rays_d']
        n_sample = self.get_ray_cfgs('n_sample')
        output = {}
        zvals, radius = self.get_zvals_outside_sphere(rays_o, rays_d, inference_only)
        out_c = self._stage(self.coarse_geo_net, self.coarse_radiance_net, rays_o, rays_d, zvals, radius, n_sample, inference_only)
        weights_c = out_c['weights']
        output['coarse'] = self.output_get_progress(out_c, get_progress)
        if self.get_ray_cfgs('n_importance') > 0:
            zvals = self.upsample_zvals(zvals, weights_c, inference_only)
            out_f = self._stage(self.fine_geo_net, self.fine_radiance_net, rays_o, rays_d, zvals, None,
                                n_sample + self.get_ray_cfgs('n_importance'), inference_only)
            output['fine'] = self.output_get_progress(out_f, get_progress)
        return self.adjust_coarse_fine_output(output, inference_only)

    def upsample_zvals(self, zvals, weights, inference_only=True):
        """coarse weights[1:n-1] on the shell mid-points -> n_importance inverse-CDF samples, merged and sorted (:117-136)"""
        w = weights[:, 1:self.get_ray_cfgs('n_sample') - 1]
        mids = 0.5 * (zvals[..., 1:] + zvals[..., :-1])
        det = True if inference_only else (not self.get_ray_cfgs('perturb'))
        new = sample_pdf(mids.contiguous(), w.detach().contiguous(), self.get_ray_cfgs('n_importance'), det).detach()
        return torch.sort(torch.cat([zvals, new], -1), -1)[0]
