"""SdfModel (arcnerf/models/sdf_model.py:11-101): foreground models whose geometry value is a signed distance (NeuS, VolSDF).
Points are evaluated with the geometry net's input gradient (the normal), which also feeds the radiance net; surface_render
(:116-185) finds the zero level by sphere tracing or the secant search and also returns the normal."""
import torch

from ..geometry.ray import get_ray_points_by_zvals, normalize
from ..utils.torch_utils import chunk_processing
from .fg_model import FgModel
from .surface_render import render_surface


class SdfModel(FgModel):
    @staticmethod
    def sigma_reverse():
        """sdf: smaller inside the object"""
        return True

    def get_est_opacity(self, dt, pts):
        raise NotImplementedError('You must implement the function in sdf-like models')

    def surface_render(self, inputs, method='sphere_tracing', n_step=128, n_iter=20, threshold=0.01, level=0.0, grad_dir='ascent'):
        assert level == 0.0, 'Invalid level for sdf model...'
        assert grad_dir == 'ascent', 'Invalid grad_dir for sdf model...'
        return render_surface(self, inputs, method, n_step, n_iter, threshold, level, grad_dir, with_normal=True)

    def field_with_normal(self, geo_net, radiance_net, pts, dirs):
        """(sdf (n), radiance (n, 3), normal (n, 3)) in chunks of chunk_pts: the sdf net and its input gradient as one first-order node
        where the net is the frequency-encoded softplus stack (ops.sdf_chain), the radiance net on top of it per chunk; else
        _forward_pts_dir per chunk as the reference has it (sdf_model.py:42-101)"""
        from ..ops.sdf_chain import sdf_chain
        res = sdf_chain(geo_net, pts, self.chunk_pts)
        if res is None:
            return chunk_processing(self._forward_pts_dir, self.chunk_pts, False, geo_net, radiance_net, pts, dirs)
        sdf, feature, normal = res
        from ..ops.radiance_chain import radiance_chain
        radiance = radiance_chain(radiance_net, pts, dirs, normal, feature, self.chunk_pts)
        if radiance is None:
            radiance = chunk_processing(radiance_net, self.chunk_pts, False, pts, dirs, normal, feature)
        return sdf[..., 0].contiguous(), radiance, normal

    def forward_pts_dir(self, pts, view_dir=None):
        geo_net, radiance_net = self.get_net()
        rays_d = torch.zeros_like(pts) if view_dir is None else normalize(view_dir)
        sigma, rgb, _ = self.field_with_normal(geo_net, radiance_net, pts, rays_d)
        return sigma, rgb

    def get_sdf_radiance_normal_by_mask_pts(self, geo_net, radiance_net, rays_o, rays_d, zvals, mask_pts=None, inference_only=False):
        """-> sdf (B,N), radiance (B,N,3), normal (B,N,3); only the valid points are evaluated, the padded tail repeats the last
        valid point's values (sdf_model.py:42-101)"""
        n_rays, n_pts = zvals.shape
        pts = get_ray_points_by_zvals(rays_o, rays_d, zvals)
        if mask_pts is None:
            pts, dirs = pts.view(-1, 3), torch.repeat_interleave(rays_d.unsqueeze(1), n_pts, dim=1).view(-1, 3)
        else:
            # one nonzero (one host sync) serves the two gathers here and the three scatters below
            flat = mask_pts.reshape(-1).nonzero(as_tuple=True)[0]
            pts = pts.reshape(-1, 3).index_select(0, flat)
            dirs = rays_d.index_select(0, torch.div(flat, n_pts, rounding_mode='floor'))
            if not inference_only:
                self.adjust_dynamicbs_factor(mask_pts)
        _sdf, _radiance, _normal = self.field_with_normal(geo_net, radiance_net, pts, dirs)
        if mask_pts is None:
            return _sdf.view(n_rays, -1), _radiance.view(n_rays, -1, 3), _normal.view(n_rays, -1, 3)
        last = torch.cumsum(mask_pts.sum(dim=1), dim=0) - 1
        sdf = torch.ones((n_rays, n_pts), dtype=zvals.dtype, device=zvals.device) * _sdf[last].unsqueeze(1)
        radiance = torch.ones((n_rays, n_pts, 3), dtype=zvals.dtype, device=zvals.device) * _radiance[last].unsqueeze(1)
        normal = torch.ones((n_rays, n_pts, 3), dtype=zvals.dtype, device=zvals.device) * _normal[last].unsqueeze(1)
        sdf = sdf.view(-1).index_copy(0, flat, _sdf.reshape(-1)).view(n_rays, n_pts)
        radiance = radiance.view(-1, 3).index_copy(0, flat, _radiance.reshape(-1, 3)).view(n_rays, n_pts, 3)
        normal = normal.view(-1, 3).index_copy(0, flat, _normal.reshape(-1, 3)).view(n_rays, n_pts, 3)
        return sdf, radiance, normal

    @staticmethod
    def _forward_pts_dir(geo_net, radiance_net, pts, rays_d=None):
        sdf, feature, normal = geo_net.forward_with_grad(pts)
        return sdf[..., 0], radiance_net(pts, rays_d, normal, feature), normal
