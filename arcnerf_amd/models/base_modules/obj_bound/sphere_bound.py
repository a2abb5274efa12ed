"""SphereBound (arcnerf/models/base_modules/obj_bound/sphere_bound.py:10-37): the object lives inside a sphere; near / far come from
the ray / sphere kernel, the zvals are BasicBound's uniform ones.  First piece of the NeuS row.

Written as a primitive-agnostic bound: a subclass names the config key that holds the primitive's constructor arguments, the
primitive class and its ray-test method; the sphere is the only primitive the reference has."""
from ....geometry.sphere import Sphere
from ....utils.cfgs_utils import valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound


class _PrimitiveBound(BasicBound):
    cfg_key = None        # field of model.obj_bound with the primitive's constructor arguments
    primitive = None      # class of the bounding primitive
    ray_test = None       # name of its method (rays_o, rays_d) -> near, far, pts, mask, each with one column per primitive

    def __init__(self, cfgs):
        assert valid_key_in_cfgs(cfgs, self.cfg_key), 'You must have {} in the cfgs'.format(self.cfg_key)
        super().__init__(cfgs)
        setattr(self, self.cfg_key, self.primitive(**vars(getattr(cfgs, self.cfg_key))))

    def get_obj_bound(self):
        return getattr(self, self.cfg_key)

    def get_near_far_from_rays(self, inputs, **kwargs):
        """-> near, far (B,1); hit (B,) bool"""
        near, far, _, hit = getattr(self.get_obj_bound(), self.ray_test)(inputs['rays_o'], inputs['rays_d'])
        return near, far, hit[:, 0]


@BOUND_REGISTRY.register()
class SphereBound(_PrimitiveBound):
    """`self.sphere` (a geometry.Sphere) bounds the object; rays starting inside it always hit"""
    cfg_key, primitive, ray_test = 'sphere', Sphere, 'ray_sphere_intersection'
