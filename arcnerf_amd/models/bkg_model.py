"""BkgModel (arcnerf/models/bkg_model.py:9-75): base of the background models - samples outside the foreground's bounding
sphere and carries the optimisation settings some backgrounds (multivol) use.  SURVEY.md section 8f, rank 2."""
import torch

from ..render.ray_helper import get_zvals_outside_sphere
from ..utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs
from .base_3d_model import Base3dModel


class BkgModel(Base3dModel):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        self.optim_cfgs = self.read_optim_cfgs()

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        raise NotImplementedError('Please implement the forward func...')

    def get_zvals_outside_sphere(self, rays_o, rays_d, inference_only=False):
        """-> zvals (B, n_sample) on the sphere shells, radius (B, n_sample, 1) of each shell"""
        zvals, radius = get_zvals_outside_sphere(rays_o, rays_d, self.get_ray_cfgs('n_sample'), self.get_ray_cfgs('bounding_radius'),
                                                 perturb=self.get_ray_cfgs('perturb') if not inference_only else False)
        return zvals, torch.repeat_interleave(radius.unsqueeze(0).unsqueeze(-1), rays_o.shape[0], 0)

    def read_optim_cfgs(self):
        optim = self.cfgs.model.optim if valid_key_in_cfgs(self.cfgs.model, 'optim') else None
        return {'near_distance': get_value_from_cfgs_field(optim, 'near_distance', 0.0),
                'epoch_optim': get_value_from_cfgs_field(optim, 'epoch_optim', 16),
                'epoch_optim_warmup': get_value_from_cfgs_field(optim, 'epoch_optim_warmup', 256),
                'ema_optim_decay': get_value_from_cfgs_field(optim, 'ema_optim_decay', 0.95),
                'opa_thres': get_value_from_cfgs_field(optim, 'opa_thres', 0.01)}

    def get_optim_cfgs(self, key=None):
        return self.optim_cfgs if key is None else self.optim_cfgs[key]

    def set_optim_cfgs(self, key, value):
        self.optim_cfgs[key] = value

    def optimize(self, cur_epoch=0):
        return
