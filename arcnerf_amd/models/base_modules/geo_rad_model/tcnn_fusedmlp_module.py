"""FusedMLPGeoNet / FusedMLPRadianceNet (arcnerf/models/base_modules/geo_rad_model/tcnn_fusedmlp_module.py:15-213) on
the hand-written f32-MFMA fused MLP kernel instead of tinycudann: no bias, hidden width in {16,32,64,128}, the geo net's
feature is its WHOLE output (column 0 doubles as the density pre-activation), fp32 end to end."""
import math

import torch
import torch.nn as nn

from .... import _native as N
from ....ops.autograd import FusedMlpFn
from ....utils.registry import MODULE_REGISTRY
from .encoder_mlp_network import EncoderMLPGeoNet, EncoderMLPRadainceNet


def get_tcnn_activation_from_cfgs(cfg, default='None'):
    if cfg is None:
        return default
    names = {'relu': 'ReLU', 'exponential': 'Exponential', 'sine': 'Sine', 'sigmoid': 'Sigmoid', 'squareplus': 'Squareplus',
             'softplus': 'Softplus'}
    if cfg.type.lower() not in names:
        raise NotImplementedError('No activation class {} in TinyCudaNN'.format(cfg.type))
    return names[cfg.type.lower()]


# tiny-cuda-nn's activation names -> the kernels' table.  Sine: forward / inference only in a fused net (its derivative needs the pre-activation,
# which neither this fused MLP nor tiny-cuda-nn's keeps); the backward entry points refuse it with that message
_KERNEL_ACT = {'ReLU': 'relu', 'None': None, 'Sigmoid': 'sigmoid', 'Exponential': 'truncexp', 'Softplus': 'softplus', 'Squareplus': 'squareplus',
               'Sine': 'sine'}


class FusedLayers(nn.Module):
    """The `layers` attribute: flat `params` like tcnn.Network, torch.nn.Linear default initialisation per layer."""

    def __init__(self, dims, activation, output_activation):
        super().__init__()
        if activation not in _KERNEL_ACT or output_activation not in _KERNEL_ACT:
            raise NotImplementedError('activation {}/{} is not provided by the fused HIP MLP'.format(activation, output_activation))
        self.dims = list(dims)
        self.desc = N.make_mlp_desc(self.dims, _KERNEL_ACT[activation], _KERNEL_ACT[output_activation], has_bias=False)
        ws = []
        for i in range(len(dims) - 1):
            w = torch.empty(dims[i + 1], dims[i])
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            ws.append(w.view(-1))
        self.params = nn.Parameter(torch.cat(ws))

    def forward(self, x):
        return FusedMlpFn.apply(x, self.params, None, self.desc)


@MODULE_REGISTRY.register()
class FusedMLPGeoNet(EncoderMLPGeoNet):
    def __init__(self, W=128, D=8, encoder=None, W_feat=128, act_cfg=None, out_act_cfg=None, dtype=torch.float32, *args,
                 **kwargs):
        super().__init__(W_feat=W_feat, out_act_cfg=out_act_cfg)
        self.W, self.D, self.dtype = W, D, dtype
        self.build_encoder(encoder)
        if W_feat > 0:
            assert W_feat in [8, 16, 32, 64, 128], 'Restrict num of layers for fused mlp.'
        n_out = W_feat if W_feat > 0 else 1
        self.layers = FusedLayers([self.embed_dim] + [W] * D + [n_out], get_tcnn_activation_from_cfgs(act_cfg, 'ReLU'), 'None')

    def forward(self, x):
        out = self.layers(self.embed_fn(x)).type(self.dtype)
        return self.handle_output_combine(out)

    def handle_output_combine(self, out):
        geo = out if self.W_feat <= 0 else out[:, 0].unsqueeze(-1)
        feat = None if self.W_feat <= 0 else out
        if self.out_act is not None:
            geo = self.out_act(geo)
        return geo, feat


@MODULE_REGISTRY.register()
class FusedMLPRadianceNet(EncoderMLPRadainceNet):
    def __init__(self, mode='vf', W=128, D=8, encoder=None, W_feat_in=128, act_cfg=None, out_act_cfg=None,
                 dtype=torch.float32, *args, **kwargs):
        super().__init__(mode=mode)
        self.W, self.D, self.W_feat_in, self.dtype = W, D, W_feat_in, dtype
        self.build_encoder(encoder, W_feat_in)
        self.layers = FusedLayers([self.init_input_dim] + [W] * D + [3], get_tcnn_activation_from_cfgs(act_cfg, 'ReLU'),
                                  get_tcnn_activation_from_cfgs(out_act_cfg, 'Sigmoid'))

    def forward(self, x, view_dirs, normals, geo_feat):
        return self.layers(self.fuse_radiance_inputs(x, view_dirs, normals, geo_feat)).type(self.dtype)
