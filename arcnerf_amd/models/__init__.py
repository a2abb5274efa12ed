"""build_model (arcnerf/models/__init__.py:21-47): `model.type` selects the foreground model class from the registry."""
from copy import deepcopy

from ..utils.cfgs_utils import valid_key_in_cfgs
from ..utils.registry import MODEL_REGISTRY
from .full_model import FullModel
from .hdrnerf_model import HDRNeRF
from .nerf_model import NeRF
from .neus_model import Neus

__all__ = ['build_model', 'FullModel', 'HDRNeRF', 'NeRF', 'Neus']


def build_model(cfgs, logger=None):
    cfgs = deepcopy(cfgs)
    fg_model = MODEL_REGISTRY.get(cfgs.model.type)(cfgs)
    if valid_key_in_cfgs(cfgs.model, 'background'):
        raise NotImplementedError('background models are not on this path yet (SURVEY.md §8f row 2)')
    model = FullModel(cfgs, fg_model)
    if logger is not None:
        logger.add_log('Model type : {}'.format(cfgs.model.type))
    return model
