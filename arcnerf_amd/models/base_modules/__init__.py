"""build_geo_model / build_radiance_model (arcnerf/models/base_modules/__init__.py:28-61)."""
from copy import deepcopy

from ...utils.cfgs_utils import valid_key_in_cfgs
from ...utils.registry import MODULE_REGISTRY
from .activation import Sine, get_activation
from .encoding import FreqEmbedder, build_encoder
from .geo_rad_model import FusedMLPGeoNet, FusedMLPRadianceNet, GeoNet, RadianceNet
from .linear import DenseLayer, SirenLayer

__all__ = ['get_activation', 'Sine', 'build_encoder', 'FreqEmbedder', 'DenseLayer', 'SirenLayer', 'build_geo_model',
           'build_radiance_model', 'GeoNet', 'RadianceNet', 'FusedMLPGeoNet', 'FusedMLPRadianceNet']


def build_geo_model(cfgs):
    cfgs = deepcopy(cfgs)
    name = cfgs.type if valid_key_in_cfgs(cfgs, 'type') else 'GeoNet'
    return MODULE_REGISTRY.get(name)(**cfgs.__dict__)


def build_radiance_model(cfgs):
    cfgs = deepcopy(cfgs)
    name = cfgs.type if valid_key_in_cfgs(cfgs, 'type') else 'RadianceNet'
    return MODULE_REGISTRY.get(name)(**cfgs.__dict__)
