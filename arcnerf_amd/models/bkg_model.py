"""BkgModel (arcnerf/models/bkg_model.py:9-75): base of the background models — sphere-shell sampling outside the foreground's
bounding sphere (NeRF++) and the refresh settings the pruned backgrounds (MultiVol) read.  SURVEY.md section 8f, rank 2."""
from ..render.ray_helper import get_zvals_outside_sphere
from ..utils.cfgs_utils import valid_key_in_cfgs
from ..utils.optim_cfgs import OptimCfgAccess, read_prune_settings
from .base_3d_model import Base3dModel

# defaults of model.optim (a background that prunes must refresh: epoch_optim is never None here)
BKG_OPTIM_DEFAULTS = {'near_distance': 0.0, 'epoch_optim': 16, 'epoch_optim_warmup': 256, 'ema_optim_decay': 0.95, 'opa_thres': 0.01}


class BkgModel(Base3dModel, OptimCfgAccess):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        self.optim_cfgs = self.read_optim_cfgs()

    def read_optim_cfgs(self):
        node = self.cfgs.model.optim if valid_key_in_cfgs(self.cfgs.model, 'optim') else None
        return read_prune_settings(node, BKG_OPTIM_DEFAULTS)

    def forward(self, inputs, inference_only=False, get_progress=False, cur_epoch=0, total_epoch=300000):
        raise NotImplementedError('Please implement the forward func...')

    def optimize(self, cur_epoch=0):
        """nothing to refresh by default"""

    def get_zvals_outside_sphere(self, rays_o, rays_d, inference_only=False):
        """-> zvals (B, n_sample) where the rays cross the shells outside `bounding_radius`, radius (B, n_sample, 1) of each"""
        jitter = bool(self.get_ray_cfgs('perturb')) and not inference_only
        zvals, shell_r = get_zvals_outside_sphere(rays_o, rays_d, self.get_ray_cfgs('n_sample'), self.get_ray_cfgs('bounding_radius'),
                                                  perturb=jitter)
        return zvals, shell_r.view(1, -1, 1).expand(rays_o.shape[0], -1, 1).contiguous()
