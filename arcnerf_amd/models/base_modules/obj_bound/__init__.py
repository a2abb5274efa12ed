"""build_obj_bound (arcnerf/models/base_modules/obj_bound/__init__.py:25-62): volume > sphere > bitfield > none."""
from copy import deepcopy

from ....utils.cfgs_utils import get_value_from_cfgs_field, valid_key_in_cfgs
from ....utils.registry import BOUND_REGISTRY
from .basic_bound import BasicBound
from .bitfield_bound import BitfieldBound
from .sphere_bound import SphereBound
from .volume_bound import VolumeBound

__all__ = ['BasicBound', 'BitfieldBound', 'SphereBound', 'VolumeBound', 'build_obj_bound']


def build_obj_bound(cfgs):
    if not valid_key_in_cfgs(cfgs, 'obj_bound'):
        return BOUND_REGISTRY.get('BasicBound')(None), 'basic'
    keys = get_value_from_cfgs_field(cfgs, 'obj_bound').__dict__.keys()
    if 'volume' in keys:
        return BOUND_REGISTRY.get('VolumeBound')(deepcopy(cfgs.obj_bound)), 'volume'
    if 'sphere' in keys:
        return BOUND_REGISTRY.get('SphereBound')(deepcopy(cfgs.obj_bound)), 'sphere'
    if 'bitfield' in keys:
        return BOUND_REGISTRY.get('BitfieldBound')(deepcopy(cfgs.obj_bound)), 'bitfield'
    raise NotImplementedError('Not such bounding class {}...'.format(list(keys)))
