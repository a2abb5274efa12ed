"""build_model (arcnerf/models/__init__.py:21-47): `model.type` selects the foreground model class from the registry."""
from copy import deepcopy

from ..utils.cfgs_utils import dict_to_obj, obj_to_dict, valid_key_in_cfgs
from ..utils.registry import MODEL_REGISTRY
from .full_model import FullModel
from .hdrnerf_model import HDRNeRF
from .multivol_bkg_model import MultiVol
from .nerf_model import NeRF
from .nerfpp_bkg_model import NeRFPP
from .neus_model import Neus

__all__ = ['build_model', 'FullModel', 'HDRNeRF', 'MultiVol', 'NeRF', 'NeRFPP', 'Neus']


def build_model(cfgs, logger=None):
    cfgs = deepcopy(cfgs)
    fg_model = MODEL_REGISTRY.get(cfgs.model.type)(cfgs)
    bkg_cfgs, bkg_model = None, None
    if valid_key_in_cfgs(cfgs.model, 'background'):
        bkg_cfgs = dict_to_obj({'model': obj_to_dict(cfgs.model.background)})
        if bkg_cfgs.model.type not in MODEL_REGISTRY:
            raise NotImplementedError('background model {} is not built yet (SURVEY.md §8f row 2)'.format(bkg_cfgs.model.type))
        bkg_model = MODEL_REGISTRY.get(bkg_cfgs.model.type)(bkg_cfgs)
    model = FullModel(cfgs, fg_model, bkg_cfgs, bkg_model)
    if logger is not None:
        logger.add_log('Model type : {}'.format(cfgs.model.type))
    return model
