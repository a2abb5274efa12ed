"""get_activation (arcnerf/models/base_modules/activation.py:24-50): config `type` -> activation module, table driven."""
import torch
import torch.nn as nn

from ...ops.trunc_exp import TruncExp
from ...utils.cfgs_utils import get_value_from_cfgs_field as _opt


class Sine(nn.Module):
    """sin(w0 x), the SIREN activation"""

    def __init__(self, w0=30.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


# type (lower case) -> constructor taking the config node; defaults are the reference's
_ACTIVATIONS = {
    'relu': lambda c: nn.ReLU(inplace=True),
    'leakyrelu': lambda c: nn.LeakyReLU(negative_slope=_opt(c, 'slope', 0.01), inplace=True),
    'softplus': lambda c: nn.Softplus(beta=_opt(c, 'beta', 100)),
    'sigmoid': lambda c: nn.Sigmoid(),
    'sine': lambda c: Sine(w0=_opt(c, 'w', 30)),
    'truncexp': lambda c: TruncExp(_opt(c, 'clip', 15.0)),
    'identity': lambda c: nn.Identity(),
}


def get_activation(cfg, default=None):
    """cfg.type selects the module; cfg None -> `default` (ReLU when that is None too)"""
    cfg = cfg if cfg is not None else default
    if cfg is None:
        return _ACTIVATIONS['relu'](None)
    make = _ACTIVATIONS.get(cfg.type.lower())
    if make is None:
        raise NotImplementedError('No activation class {}'.format(cfg.type))
    return make(cfg)
