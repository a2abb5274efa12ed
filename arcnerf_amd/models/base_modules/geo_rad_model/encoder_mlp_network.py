"""Encoder + MLP scaffolding shared by the linear and the fused nets
(arcnerf/models/base_modules/geo_rad_model/encoder_mlp_network.py:11-118)."""
import torch

from ....geometry.ray import normalize
from ..activation import get_activation
from ..encoding import build_encoder
from .base_network import BaseGeoNet, BaseRadianceNet


class EncoderMLPGeoNet(BaseGeoNet):
    def __init__(self, W_feat=256, out_act_cfg=None):
        super().__init__()
        self.W_feat = W_feat
        self.embed_dim, self.embed_fn = 0, None
        self.out_act = get_activation(cfg=out_act_cfg) if out_act_cfg is not None else None

    def build_encoder(self, encoder):
        self.embed_fn, input_ch, embed_freq = build_encoder(encoder)
        self.embed_dim = self.embed_fn.get_output_dim()
        return input_ch, embed_freq

    def handle_output(self, out):
        """linear nets: out = [geo | feat]"""
        if self.W_feat <= 0:
            geo, feat = out, None
        elif out.shape[-1] > 1 + self.W_feat:
            # the final product's own padded tensor (linear(keep_pad=True)): ONE split node, whose backward is one concatenation of the
            # two gradients and the zero pad instead of two slice nodes (zero fill + copy each) and an add
            geo, feat, _ = torch.split(out, [1, self.W_feat, out.shape[-1] - 1 - self.W_feat], dim=-1)
        else:
            geo, feat = out[:, 0].unsqueeze(-1), out[:, 1:]
        if self.out_act is not None:
            geo = self.out_act(geo)
        return geo, feat


class EncoderMLPRadainceNet(BaseRadianceNet):
    """`mode` is a string over {p: encoded position, v: encoded unit view direction, n: normal, f: geometry feature}; the net's
    input is those blocks concatenated in the order the characters appear (encoder_mlp_network.py:62-118).  (The class name keeps
    the reference's spelling: it is part of the interface.)"""

    def __init__(self, mode='vf'):
        super().__init__()
        assert len(mode) > 0 and set(mode) <= set('pvnf'), 'Invalid mode only pvnf allowed...'
        self.mode = mode
        self.init_input_dim = 0
        self.embed_fn_pts = self.embed_fn_view = None

    def build_encoder(self, encoder, W_feat_in):
        """encoders for the p / v blocks and the total input width"""
        width = {'n': 3, 'f': max(int(W_feat_in), 0)}
        if 'p' in self.mode:
            self.embed_fn_pts = build_encoder(getattr(encoder, 'pts', None) if encoder is not None else None)[0]
            width['p'] = self.embed_fn_pts.get_output_dim()
        if 'v' in self.mode:
            self.embed_fn_view = build_encoder(getattr(encoder, 'view', None) if encoder is not None else None)[0]
            width['v'] = self.embed_fn_view.get_output_dim()
        self.init_input_dim = sum(width[m] for m in self.mode)

    def fuse_radiance_inputs(self, x, view_dirs, normals, geo_feat, pad4=False):
        block = {'p': lambda: self.embed_fn_pts(x), 'v': lambda: self.embed_fn_view(normalize(view_dirs)),
                 'n': lambda: normals, 'f': lambda: geo_feat}
        parts = [block[m]() for m in self.mode]
        assert sum(p.shape[-1] for p in parts) == self.init_input_dim, 'Shape not match'
        kp = (-self.init_input_dim) % 4
        if pad4 and kp and parts[0].is_cuda and parts[0].dtype == torch.float32:
            # the first layer's padded input width (283 -> 284) from the concat itself, not a second copy
            parts.append(parts[0].new_zeros(parts[0].shape[:-1] + (kp,)))
        return torch.cat(parts, dim=-1)
