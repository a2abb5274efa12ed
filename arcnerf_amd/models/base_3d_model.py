"""Base3dModel (arcnerf/models/base_3d_model.py:14-389): ray config, chunk sizes, ray marching wrapper, point queries."""
import torch
import torch.nn as nn

from ..geometry.ray import normalize
from ..render.ray_helper import ray_marching
from ..utils.cfgs_utils import get_value_from_cfgs_field
from ..utils.torch_utils import chunk_processing


class Base3dModel(nn.Module):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        r = cfgs.model.rays
        g = get_value_from_cfgs_field
        self.ray_cfgs = {
            'bounding_radius': g(r, 'bounding_radius'), 'volume': g(r, 'volume'), 'near': g(r, 'near'), 'far': g(r, 'far'),
            'n_sample': g(r, 'n_sample', 128), 'inverse_linear': g(r, 'inverse_linear', False), 'perturb': g(r, 'perturb', False),
            'add_inf_z': g(r, 'add_inf_z', False), 'noise_std': g(r, 'noise_std', False), 'white_bkg': g(r, 'white_bkg', False),
            'rand_bkg_color': g(r, 'rand_bkg_color', False),
        }
        self.chunk_rays = cfgs.model.chunk_rays
        self.chunk_pts = cfgs.model.chunk_pts
        self.add_inf_z = self.ray_cfgs['add_inf_z']

    def set_add_inf_z(self, v):
        self.add_inf_z = v

    @staticmethod
    def sigma_reverse():
        return False

    def get_ray_cfgs(self, key=None):
        return self.ray_cfgs if key is None else self.ray_cfgs[key]

    def set_ray_cfgs(self, key, value):
        self.ray_cfgs[key] = value

    def get_chunk_rays(self):
        return self.chunk_rays

    def get_chunk_pts(self):
        return self.chunk_pts

    def set_chunk_rays(self, v):
        self.chunk_rays = v

    def set_chunk_pts(self, v):
        self.chunk_pts = v

    def get_net(self):
        return self.geo_net, self.radiance_net

    def init_setting(self):
        self.get_net()[0].pretrain_siren()

    def get_obj_bound_type(self):
        return None

    def get_obj_bound_structure(self):
        return None

    def get_dynamicbs_factor(self):
        return 1

    def reset_measurement(self):
        return

    def ray_marching(self, sigma, radiance, zvals, add_inf_z=None, alpha=None, inference_only=False, weights_only=False,
                     bkg_color=None):
        noise_std = self.get_ray_cfgs('noise_std') if not inference_only else 0.0
        return ray_marching(sigma, radiance, zvals, self.add_inf_z if add_inf_z is None else add_inf_z, float(noise_std or 0.0),
                            weights_only=weights_only, white_bkg=self.get_ray_cfgs('white_bkg'), alpha=alpha, bkg_color=bkg_color)

    def output_get_progress(self, output, get_progress=False, n_fg=None):
        keys = ['sigma', 'zvals', 'alpha', 'trans_shift', 'weights', 'radiance']
        if get_progress:
            for k in keys:
                output['progress_' + k] = output[k] if n_fg is None else output[k][:, :n_fg]
            if self.sigma_reverse():
                output['progress_sigma_reverse'] = True
        for k in keys:
            output.pop(k)
        return output

    def adjust_coarse_fine_output(self, output, inference_only=False):
        assert 'n_importance' in self.get_ray_cfgs(), 'Not valid for two stage model...'
        has_fine = self.get_ray_cfgs('n_importance') > 0
        if inference_only:
            return output['fine'] if has_fine else output['coarse']
        out = {k + '_coarse': v for k, v in output['coarse'].items()}
        if has_fine:
            out.update({k + '_fine': v for k, v in output['fine'].items()})
        return out

    @staticmethod
    def _forward_pts_dir(geo_net, radiance_net, pts, rays_d=None):
        sigma, feature = geo_net(pts)
        return sigma[..., 0], radiance_net(pts, rays_d, None, feature)

    def field_on_points(self, geo_net, radiance_net, pts, dirs):
        """(sigma (n), radiance (n, 3)) for points / directions in chunks of chunk_pts: one autograd node where the two nets are the
        plain frequency-encoded ReLU stacks (ops.field_chain), else _forward_pts_dir per chunk as the reference has it
        (base_3d_model.py:335-366)"""
        from ..ops.field_chain import field_chain
        out = field_chain(geo_net, radiance_net, pts, dirs, self.chunk_pts)
        if out is not None:
            return out
        return chunk_processing(self._forward_pts_dir, self.chunk_pts, False, geo_net, radiance_net, pts, dirs)

    def forward_pts_dir(self, pts, view_dir=None):
        geo_net, radiance_net = self.get_net()
        rays_d = torch.zeros_like(pts) if view_dir is None else normalize(view_dir)
        return self.field_on_points(geo_net, radiance_net, pts, rays_d)

    def forward_pts(self, pts):
        geo_net, _ = self.get_net()
        return chunk_processing(geo_net.forward_geo_value, self.chunk_pts, False, pts)

    def get_est_opacity(self, dt, pts):
        """instant-ngp style opacity estimate sigma * dt (base_3d_model.py:368-389)"""
        return self.forward_pts(pts) * dt
