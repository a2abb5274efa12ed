"""build_encoder (arcnerf/models/base_modules/encoding/__init__.py:26-51): `type:` selects the encoder class, the whole
config block is splatted as keyword arguments; returns (module, input_dim, n_freqs)."""
from copy import deepcopy

from ....utils.cfgs_utils import dict_to_obj, valid_key_in_cfgs
from ....utils.registry import ENCODER_REGISTRY
from .freq_encoder import FreqEmbedder
from .hashgrid_encoder import HashGridEmbedder
from .sh_encoder import SHEmbedder

__all__ = ['FreqEmbedder', 'HashGridEmbedder', 'SHEmbedder', 'build_encoder']


def build_encoder(cfgs):
    if cfgs is None:
        cfgs = dict_to_obj({'type': 'FreqEmbedder', 'input_dim': 3, 'n_freqs': 0})
    cfgs = deepcopy(cfgs)
    name = cfgs.type if valid_key_in_cfgs(cfgs, 'type') else 'FreqEmbedder'
    return ENCODER_REGISTRY.get(name)(**cfgs.__dict__), cfgs.input_dim, cfgs.n_freqs
