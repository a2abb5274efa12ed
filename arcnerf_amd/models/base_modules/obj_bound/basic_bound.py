"""BasicBound (arcnerf/models/base_modules/obj_bound/basic_bound.py:12-106): no structure, near/far from config or data."""
import torch
import torch.nn as nn

from ....render.ray_helper import get_near_far_from_rays, get_zvals_from_near_far
from ....utils.optim_cfgs import OptimCfgAccess, read_prune_settings
from ....utils.registry import BOUND_REGISTRY

# defaults of the refresh settings under model.obj_bound (no `epoch_optim` = the structure is never pruned)
BOUND_OPTIM_DEFAULTS = {'epoch_optim': None, 'epoch_optim_warmup': None, 'ema_optim_decay': 0.95, 'opa_thres': 0.01}


@BOUND_REGISTRY.register()
class BasicBound(nn.Module, OptimCfgAccess):
    def __init__(self, cfgs):
        super().__init__()
        self.cfgs = cfgs
        self.optim_cfgs = self.read_optim_cfgs()

    def get_obj_bound(self):
        return None

    def read_optim_cfgs(self):
        return read_prune_settings(self.cfgs, BOUND_OPTIM_DEFAULTS)

    def get_near_far_from_rays(self, inputs, near_hardcode=None, far_hardcode=None, bounding_radius=None):
        """-> near, far (B,1), mask_rays None"""
        near, far = get_near_far_from_rays(inputs['rays_o'], inputs['rays_d'], inputs.get('bounds'), near_hardcode,
                                           far_hardcode, bounding_radius)
        return near, far, None

    def get_zvals_from_near_far(self, near, far, n_pts, inference_only=False, inverse_linear=False, perturb=False, **kwargs):
        """-> zvals (B,n_pts), mask_pts None"""
        return get_zvals_from_near_far(near, far, n_pts, inverse_linear=inverse_linear,
                                       perturb=perturb if not inference_only else False), None

    @torch.no_grad()
    def optimize(self, cur_epoch=0, n_pts=128, get_est_opacity=None):
        return
