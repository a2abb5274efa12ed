"""get_activation (arcnerf/models/base_modules/activation.py:24-50): config `type` -> activation module."""
import torch
import torch.nn as nn

from ...ops.trunc_exp import TruncExp
from ...utils.cfgs_utils import dict_to_obj, get_value_from_cfgs_field


class Sine(nn.Module):
    def __init__(self, w0=30.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):
        return torch.sin(self.w0 * x)


def get_activation(cfg, default=None):
    if cfg is None:
        cfg = default if default is not None else dict_to_obj({'type': 'relu'})
    kind = cfg.type.lower()
    if kind == 'relu':
        return nn.ReLU(inplace=True)
    if kind == 'softplus':
        return nn.Softplus(beta=get_value_from_cfgs_field(cfg, 'beta', 100))
    if kind == 'leakyrelu':
        return nn.LeakyReLU(negative_slope=get_value_from_cfgs_field(cfg, 'slope', 0.01), inplace=True)
    if kind == 'sine':
        return Sine(w0=get_value_from_cfgs_field(cfg, 'w', 30))
    if kind == 'sigmoid':
        return nn.Sigmoid()
    if kind == 'truncexp':
        return TruncExp(get_value_from_cfgs_field(cfg, 'clip', 15.0))
    if kind == 'identity':
        return nn.Identity()
    raise NotImplementedError('No activation class {}'.format(cfg.type))
