"""SphereBound (arcnerf/models/base_modules/obj_bound/sphere_bound.py:10-37): near/far from the ray / sphere kernel, uniform
zvals inherited from BasicBound.  First piece of the NeuS row."""
from ....geometry.sphere import Sphere
from ....utils.cfgs_utils import valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


@BOUND_REGISTRY.register()
class SphereBound(BasicBound):
    def __init__(self, cfgs):
        assert valid_key_in_cfgs(cfgs, 'sphere'), 'You must have sphere in the cfgs'
        super().__init__(cfgs)
        self.sphere = Sphere(**vars(cfgs.sphere))

    def get_obj_bound(self):
        return self.sphere

    def get_near_far_from_rays(self, inputs, **kwargs):
        """-> near, far (B,1); hit (B,) bool: rays that meet the sphere (always true from inside)"""
        near, far, _, hit = self.sphere.ray_sphere_intersection(inputs['rays_o'], inputs['rays_d'])
        return near, far, hit.squeeze(-1)
