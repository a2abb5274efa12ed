"""SphereBound (arcnerf/models/base_modules/obj_bound/sphere_bound.py:10-37): near/far from the ray / sphere test, uniform
zvals from BasicBound.  First piece of the NeuS row."""
from ....geometry.sphere import Sphere
from ....utils.cfgs_utils import valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


@BOUND_REGISTRY.register()
class SphereBound(BasicBound):
    def __init__(self, cfgs):
        super().__init__(cfgs)
        assert valid_key_in_cfgs(cfgs, 'sphere'), 'You must have sphere in the cfgs'
        self.sphere = Sphere(**cfgs.sphere.__dict__)

    def get_obj_bound(self):
        return self.sphere

    def get_near_far_from_rays(self, inputs, **kwargs):
        """-> near, far (B,1), mask_rays (B,) bool"""
        near, far, _, mask_rays = self.sphere.ray_sphere_intersection(inputs['rays_o'], inputs['rays_d'])
        return near, far, mask_rays[:, 0]
