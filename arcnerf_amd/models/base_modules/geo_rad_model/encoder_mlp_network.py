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
        else:
            geo, feat = out[:, 0].unsqueeze(-1), out[:, 1:]
        if self.out_act is not None:
            geo = self.out_act(geo)
        return geo, feat


class EncoderMLPRadainceNet(BaseRadianceNet):
    def __init__(self, mode='vf'):
        super().__init__()
        assert len(mode) > 0 and all(m in 'pvnf' for m in mode), 'Invalid mode only pvnf allowed...'
        self.mode = mode
        self.init_input_dim = 0
        self.embed_fn_pts = self.embed_fn_view = None

    def build_encoder(self, encoder, W_feat_in):
        if 'p' in self.mode:
            self.embed_fn_pts, _, _ = build_encoder(encoder.pts if encoder is not None else None)
            self.init_input_dim += self.embed_fn_pts.get_output_dim()
        if 'v' in self.mode:
            self.embed_fn_view, _, _ = build_encoder(encoder.view if encoder is not None else None)
            self.init_input_dim += self.embed_fn_view.get_output_dim()
        if 'n' in self.mode:
            self.init_input_dim += 3
        if 'f' in self.mode and W_feat_in > 0:
            self.init_input_dim += W_feat_in

    def fuse_radiance_inputs(self, x, view_dirs, normals, geo_feat):
        """inputs concatenated in the order of the characters of `mode` (encoder_mlp_network.py:93-118)"""
        parts = {}
        if 'p' in self.mode:
            parts['p'] = self.embed_fn_pts(x)
        if 'v' in self.mode:
            parts['v'] = self.embed_fn_view(normalize(view_dirs))
        if 'n' in self.mode:
            parts['n'] = normals
        if 'f' in self.mode:
            parts['f'] = geo_feat
        out = torch.cat([parts[m] for m in self.mode], dim=-1)
        assert out.shape[-1] == self.init_input_dim, 'Shape not match'
        return out
