"""GeoNet / RadianceNet: nn.Linear stacks with skip concatenation and geometric initialisation
(arcnerf/models/base_modules/geo_rad_model/linear_network_module.py:16-335).  Every layer's product, its gradients and (NeuS) their
gradients run on the hand-written f32-MFMA kernels of csrc/gemm.hip (base_modules/linear.py); parameter names
(`layers.{i}.weight/bias`, `embed_fn...`) match the reference's state_dict."""
import math

import numpy as np
import torch
import torch.nn as nn

from ....utils.cfgs_utils import dict_to_obj
from ....utils.registry import MODULE_REGISTRY
from ..activation import get_activation
from ..linear import DenseLayer, Linear, SirenLayer
from .encoder_mlp_network import EncoderMLPGeoNet, EncoderMLPRadainceNet




@MODULE_REGISTRY.register()
class GeoNet(EncoderMLPGeoNet):
    def __init__(self, W=256, D=8, skips=[4], encoder=None, W_feat=256, use_bias=True, skip_reduce_output=False,
                 norm_skip=False, act_cfg=None, geometric_init=True, radius_init=1.0, use_siren=False, weight_norm=False,
                 out_act_cfg=None, *args, **kwargs):
        super().__init__(W_feat=W_feat, out_act_cfg=out_act_cfg)
        self.W, self.D, self.skips, self.norm_skip = W, D, list(skips), norm_skip
        self.geometric_init, self.use_siren, self.radius_init = geometric_init, use_siren, radius_init
        self.is_pretrained = False
        input_ch, embed_freq = self.build_encoder(encoder)
        if use_siren:
            assert len(self.skips) == 0, 'do not use skips for siren'
        layers = []
        for i in range(D + 1):
            after_skip = i > 0 and (i - 1) in self.skips
            in_dim = self.embed_dim if i == 0 else (self.embed_dim + W if (after_skip and not skip_reduce_output) else W)
            if i == D:
                out_dim = 1 + W_feat if W_feat > 0 else 1
            elif skip_reduce_output and i in self.skips:
                out_dim = W - self.embed_dim
            else:
                out_dim = W
            if i == D:
                layer = Linear(in_dim, out_dim, bias=use_bias)
            elif use_siren:
                layer = SirenLayer(in_dim, out_dim, is_first=(i == 0), bias=use_bias)
            else:
                layer = DenseLayer(in_dim, out_dim, activation=get_activation(act_cfg), bias=use_bias)
            if geometric_init and not use_siren:
                self._geometric_init(layer, i, in_dim, out_dim, input_ch, embed_freq, use_bias, after_skip)
            if weight_norm:
                layer = nn.utils.weight_norm(layer)
            layers.append(layer)
        self.layers = nn.ModuleList(layers)

    def _geometric_init(self, layer, i, in_dim, out_dim, input_ch, embed_freq, use_bias, after_skip):
        """sphere-like sdf initialisation (linear_network_module.py:139-163); layer inputs are [feature, x, embed_x]"""
        if i == self.D:
            nn.init.normal_(layer.weight, mean=np.sqrt(np.pi) / np.sqrt(in_dim), std=0.0001)
            if use_bias:
                nn.init.constant_(layer.bias[:1], -self.radius_init)
            return
        if use_bias:
            nn.init.constant_(layer.bias, 0.0)
        std = np.sqrt(2) / np.sqrt(out_dim)
        if embed_freq > 0 and i == 0:
            nn.init.constant_(layer.weight[:, input_ch:], 0.0)
            nn.init.normal_(layer.weight[:, :input_ch], 0.0, std)
        elif embed_freq > 0 and after_skip:
            nn.init.normal_(layer.weight, 0.0, std)
            nn.init.constant_(layer.weight[:, -(self.embed_dim - input_ch):], 0.0)
        else:
            nn.init.normal_(layer.weight, 0.0, std)

    def forward_with_grad(self, x):
        """(geo, feat, d geo / d x).  For the sdf net of NeuS on the hash grid - one hidden softplus layer, no biases, nothing but the hash
        features as input - the input gradient is an explicit output of one first-order node (ops.autograd.SdfMlpJacFn) pushed through
        the encoding's d enc / d x kernel, instead of a create_graph differentiation of the layer stack; everything else takes
        base_network.py:30-44."""
        fast = self._jacobian_path(x)
        return fast if fast is not None else super().forward_with_grad(x)

    nograd_fast = True      # graph-free passes take ops.sdf_chain.sdf_forward_nograd where it applies (False: the layer-by-layer modules - the tests' reference)

    def _jacobian_path(self, x):
        from ..encoding.hashgrid_encoder import HashGridEmbedder
        if not (x.is_cuda and x.dtype == torch.float32 and x.numel() > 0):
            return None
        if x.requires_grad or self.D != 1 or self.skips or self.W_feat <= 0 or self.out_act is not None:
            return None
        emb, l0, l1 = self.embed_fn, self.layers[0], self.layers[1]
        if type(emb) is not HashGridEmbedder or emb.include_input:
            return None
        if type(l0) is not DenseLayer or type(l0.activation) is not nn.Softplus or l0.activation.threshold != 20 or type(l1) is not Linear:
            return None
        if l0.bias is not None or l1.bias is not None or hasattr(l0, 'weight_g') or hasattr(l1, 'weight_g'):
            return None
        from ....ops.autograd import HashGridDxFn, HashGridFn, SdfMlpJacFn
        pts = x.detach().contiguous()
        with torch.enable_grad():
            enc = HashGridFn.apply(pts, emb.embeddings, emb.desc, True)
            n_out = l1.weight.shape[0]
            w2 = torch.nn.functional.pad(l1.weight, (0, 0, 0, (-n_out) % 4))     # rows to a multiple of 4: aligned products both ways
            out, jac = SdfMlpJacFn.apply(enc, l0.weight, w2, float(l0.activation.beta))
            geo, feat = self.handle_output(out[:, :n_out])
            grad = HashGridDxFn.apply(pts, emb.embeddings, jac, emb.desc)
        return geo, feat, grad

    def forward(self, x):
        if not torch.is_grad_enabled() and self.nograd_fast:
            # a graph-free pass of the softplus sdf net (NeuS evaluates it in every importance-sampling round): the cached weight-normed, padded
            # weights, activations in the products' epilogues, the skip concatenation as one kernel - no hooks, pads, cat or div launches
            from ....ops.sdf_chain import sdf_forward_nograd
            r = sdf_forward_nograd(self, x)
            if r is not None:
                return r
        x_embed = self.embed_fn(x)
        if all(self.layers[i].out_features % 4 == 0 for i in self.skips if i <= self.D):
            # (63 -> 64 columns ONCE where the layers run on the HIP products: layer 0 and the skip concat both take the padded
            # block, 256 + 64 = the padded 319, instead of a pad copy of each)
            from ....ops.autograd import pad_cols4
            x_embed = pad_cols4(x_embed)
        out = x_embed
        for i in range(self.D + 1):
            if i == self.D and self.W_feat > 0 and type(self.layers[i]) is Linear and not hasattr(self.layers[i], 'weight_g'):
                from ....ops.autograd import linear
                out = linear(out, self.layers[i].weight, self.layers[i].bias, keep_pad=True)    # handle_output splits the padded tensor
            elif i in self.skips and type(self.layers[i]) is DenseLayer and type(self.layers[i].activation) is nn.ReLU and \
                    not hasattr(self.layers[i], 'weight_g'):
                # the layer in front of a skip concatenation writes its columns of the concatenated tensor itself
                from ....ops.autograd import linear_relu_cat
                cat = linear_relu_cat(out, self.layers[i].weight, self.layers[i].bias, x_embed)
                out = cat if cat is not None else torch.cat([self.layers[i](out), x_embed], dim=-1)
                if self.norm_skip:
                    out = out / math.sqrt(2)
                continue
            else:
                out = self.layers[i](out)
            if i in self.skips:
                out = torch.cat([out, x_embed], dim=-1)
                if self.norm_skip:
                    out = out / math.sqrt(2)
        return self.handle_output(out)


@MODULE_REGISTRY.register()
class RadianceNet(EncoderMLPRadainceNet):
    def __init__(self, mode='vf', W=256, D=8, encoder=None, W_feat_in=256, use_bias=True, act_cfg=None, use_siren=False,
                 weight_norm=False, out_act_cfg=None, *args, **kwargs):
        super().__init__(mode=mode)
        self.W, self.D, self.W_feat_in = W, D, W_feat_in
        self.build_encoder(encoder, W_feat_in)
        layers = []
        for i in range(D + 1):
            in_dim = self.init_input_dim if i == 0 else W
            if i == D:
                act = get_activation(out_act_cfg, dict_to_obj({'type': 'Sigmoid'}))
                layer = DenseLayer(in_dim, 3, activation=act, bias=use_bias)
            elif use_siren:
                layer = SirenLayer(in_dim, W, is_first=(i == 0), bias=use_bias)
            else:
                layer = DenseLayer(in_dim, W, activation=get_activation(act_cfg), bias=use_bias)
            if weight_norm:
                layer = nn.utils.weight_norm(layer)
            layers.append(layer)
        self.layers = nn.ModuleList(layers)
        self._fused_desc = self._fused_shape()

    def _fused_shape(self):
        """MLP descriptor of the fused HIP kernel when this stack is one it computes exactly like the nn.Linear chain: two or three
        bias-free DenseLayers (no weight norm, no SIREN), ReLU inside, sigmoid out, every width <= 64 (the reference's hash-grid
        configurations, e.g. the radiance net of NeuS-NGP).  The parameters stay the layers' own `weight`s (same state_dict); only
        the arithmetic moves from three library GEMMs + activations per direction to one kernel."""
        if len(self.layers) not in (2, 3):
            return None
        for i, layer in enumerate(self.layers):
            last = i == len(self.layers) - 1
            if type(layer) is not DenseLayer or layer.bias is not None or hasattr(layer, 'weight_g'):
                return None
            if not isinstance(layer.activation, nn.Sigmoid if last else nn.ReLU):
                return None
        dims = [self.layers[0].in_features] + [layer.out_features for layer in self.layers]
        if max(dims) > 64:
            return None
        from .... import _native as N
        return N.make_mlp_desc(dims, 'relu', 'sigmoid', has_bias=False)

    def forward(self, x, view_dirs, normals, geo_feat):
        out = self.fuse_radiance_inputs(x, view_dirs, normals, geo_feat, pad4=self._fused_desc is None)
        if self._fused_desc is not None and out.is_cuda and out.dtype == torch.float32:
            from ....ops.autograd import FusedMlpFn
            weights = torch.cat([layer.weight.reshape(-1) for layer in self.layers])
            return FusedMlpFn.apply(out.contiguous(), weights, None, self._fused_desc)
        for layer in self.layers:
            out = layer(out)
        return out
